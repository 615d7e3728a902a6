#!/bin/bash
# VGPRs / scratch / LDS / occupancy of every kernel of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
#   bash tools/kernel_resources.sh conv_bf16 [extra hipcc flags]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd/csrc
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$CS "$@" -c $CS/$f.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  sed -E 's/ \[-Rpass-analysis=kernel-resource-usage\]//' |
  awk '/Function Name:/{name=$NF} / VGPRs:/{v=$NF} /AGPRs:/{a=$NF} /TotalSGPRs/{sg=$NF} /ScratchSize/{s=$NF} /Occupancy/{o=$NF} /VGPRs Spill/{sp=$NF} /LDS Size/{print name, "vgpr", v, "agpr", a, "sgpr", sg, "scratch", s, "spill", sp, "lds", $NF, "occ", o}' |
  c++filt | sed -E 's/\(anonymous namespace\):://; s/\(rnh_[a-z0-9_]*args.*\)//'
