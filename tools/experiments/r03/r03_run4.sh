set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_d
mkdir -p $O
for rep in 1 2; do MASKS="0 32 64 128 96 160" bash tools/bf16_ablate.sh run 2>&1 | grep -v "wgrad\|gates_bwd"; done > $O/ablate2.txt
cat $O/ablate2.txt
