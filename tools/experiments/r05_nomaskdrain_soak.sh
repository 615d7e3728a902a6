# Soak of the direct implicit GEMM without its masked-lane drain (the product build since round 5; before: -DRNH_IGEMM_NO_MASK_DRAIN): cold training
# steps of the width-16 fp32 net beside the helper stream, bitwise against the first, at three image sizes; then the parity tests that run the
# implicit-GEMM kernels, with that library.  One box per call: run it in several calls.
PKG=$PWD/efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd
o=gpurun_out/${1:-r05u}; mkdir -p $o
# (since round 5 the product build IS the build without the drain; RNH_LIB selects another library)
for sz in 32 64 128; do
  reps=1500; [ $sz = 128 ] && reps=700
  echo "== size $sz, $reps cold steps" >> $o/soak.txt
  PROBE_SIZE=$sz RNH_POISON=1 timeout -k 10 500 python tools/probes/flake_width16.py $reps 16 f32 2>&1 | grep -v amdgpu | tail -3 >> $o/soak.txt
done
cat $o/soak.txt
