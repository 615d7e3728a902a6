#!/bin/bash
# Build librefinenet_hip.so (gfx950 only) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="${RNH_OUT:-$HERE/../hipvsr/librefinenet_hip.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
mkdir -p "$(dirname "$OUT")"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I"$ROOT/include" -I"$HERE" \
    "$HERE/conv_igemm.hip" "$HERE/conv_wino.hip" "$HERE/conv_wgrad.hip" "$HERE/wgrad_wino.hip" "$HERE/small_kernels.hip" "$HERE/uptail.hip" "$HERE/cine_gather.hip" "$HERE/step_tail.hip" -o "$OUT" "$@"
echo "built $OUT"
