#!/bin/bash
# Build librefinenet_hip.so (gfx950 only) in-tree.  hipcc cross-compiles without a GPU.  One object per source file,
# compiled in parallel, then one link.  Extra arguments (e.g. -DRNH_STAMPS) go to every compile.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="${RNH_OUT:-$HERE/../hipvsr/librefinenet_hip.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
OBJ="$(mktemp -d "${TMPDIR:-/tmp}/rnh_build.XXXXXX")"
trap 'rm -rf "$OBJ"' EXIT
mkdir -p "$(dirname "$OUT")"
SRCS=(conv_igemm conv_wino conv_wino44 wgrad_wino44 wgrad_wino44f conv_wgrad wgrad_wino small_kernels uptail uptail_bf16 cine_gather step_tail conv_bf16 wgrad_bf16 mixed_kernels)
# RNH_PERSISTENT=1: also build the persistent form of the bf16 convolution (csrc/experiments/conv_bf16p.hip: measured slower in round 5, kept for A/B runs)
if [ "${RNH_PERSISTENT:-0}" = 1 ]; then SRCS+=(experiments/conv_bf16p); set -- "$@" -DRNH_WITH_PERSISTENT; mkdir -p "$OBJ/experiments"; fi
pids=()
for s in "${SRCS[@]}"; do
    extra=()
    # wgrad_wino: hipcc's SLP vectoriser pairs the transform adds into v_pk_add_f32 at the price of four v_mov per pair
    # (82 moves per loop iteration of the LDS kernel); the add/subtract networks are cheaper as they are written
    [ "$s" = wgrad_wino ] && extra=(-fno-slp-vectorize)
    "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$ROOT/include" -I"$HERE" "${extra[@]}" "$@" -c "$HERE/$s.hip" -o "$OBJ/$s.o" &
    pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
(cd "$OBJ" && "$HIPCC" --offload-arch=gfx950 -shared -fPIC "${SRCS[@]/%/.o}" -o "$OUT")
echo "built $OUT"
