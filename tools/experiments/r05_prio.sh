# s_setprio of the main loop / of the epilogue of conv_bf16d_kernel: libraries lib_prio<main><epi>.so built with -DRNH_PRIO=<main> -DRNH_PRIO_EPI=<epi>, same box, alternating
PKG=$PWD/efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd
mkdir -p gpurun_out/r05d
for rep in 1 2; do for m in 00 02 03 13 20; do echo "== main/epilogue priority $m (run $rep)"; RNH_LIB=$PKG/hipvsr/lib_prio$m.so python tools/kbench_bf16.py 2>&1 | grep -v amdgpu.ids | grep -E "lstm.fwd|lstm.nog|lstm.dgrad|up1.fwd|refine1.fwd|refine2.fwd"; done; done > gpurun_out/r05d/prio_epi.txt 2>&1
cat gpurun_out/r05d/prio_epi.txt
