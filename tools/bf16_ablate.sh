# Ablation of the bf16 ConvLSTM / data-gradient kernel's main loop (diagnostic builds with -DRNH_EXP=<mask>, csrc/conv_bf16.hip):
# which of weight streaming, halo fragment reads, halo staging, the chunk barrier and the epilogue costs how much of a launch.
#   build here:  bash tools/bf16_ablate.sh build       (libraries hipvsr/lib_exp<mask>.so travel to the GPU box)
#   GPU box:     bash tools/bf16_ablate.sh run > gpurun_out/ablate.txt
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd
MASKS="${MASKS:-0 1 2 4 8 16 3 7 15 31}"
if [ "$1" = build ]; then
  for m in $MASKS; do RNH_OUT=$PKG/hipvsr/lib_exp$m.so bash $PKG/csrc/build.sh -DRNH_EXP=$m > /dev/null; done
  ls -la $PKG/hipvsr/lib_exp*.so
else
  cd $ROOT
  for m in $MASKS; do echo "== RNH_EXP=$m"; RNH_LIB=$PKG/hipvsr/lib_exp$m.so python tools/kbench_bf16.py lstm. 2>&1 | grep -v amdgpu.ids; done
fi
