set -e
PKG=$PWD/efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd
mkdir -p gpurun_out/r05c
for m in 0 16 31 256 257 260 264 272 263 271 287; do
  echo "== RNH_EXP=$m"
  if [ $m = 0 ]; then python tools/kbench_bf16.py lstm.fwd 2>&1 | grep -v amdgpu.ids; python tools/kbench_bf16.py lstm.dgrad 2>&1 | grep -v amdgpu.ids
  else RNH_LIB=$PKG/hipvsr/lib_exp$m.so python tools/kbench_bf16.py lstm.fwd 2>&1 | grep -v amdgpu.ids; RNH_LIB=$PKG/hipvsr/lib_exp$m.so python tools/kbench_bf16.py lstm.dgrad 2>&1 | grep -v amdgpu.ids; fi
done > gpurun_out/r05c/ablate_solo.txt 2>&1
echo "== stamps paired" > gpurun_out/r05c/stamps.txt
python tools/bf16_stamps.py lstm >> gpurun_out/r05c/stamps.txt 2>&1
echo "== stamps solo" >> gpurun_out/r05c/stamps.txt
RNH_LIB=$PKG/hipvsr/lib_stamps256.so python tools/bf16_stamps.py lstm >> gpurun_out/r05c/stamps.txt 2>&1
RNH_LIB=$PKG/hipvsr/lib_stamps256.so python tools/bf16_wgtrace.py lstm > gpurun_out/r05c/wgtrace_solo.txt 2>&1
cat gpurun_out/r05c/ablate_solo.txt gpurun_out/r05c/stamps.txt gpurun_out/r05c/wgtrace_solo.txt
