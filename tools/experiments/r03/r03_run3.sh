set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_c
mkdir -p $O
python tools/kbench_bf16.py > $O/kbench_all.txt 2>&1
cat $O/kbench_all.txt | grep -v amdgpu.ids
bash tools/bf16_ablate.sh run > $O/ablate.txt 2>&1
cat $O/ablate.txt
bash tools/prof_pmc_bf16.sh r03c lstm. > $O/pmc.log 2>&1 || true
cp gpurun_out/pmc_r03c/summary.json $O/pmc_summary.json || true
tail -80 $O/pmc.log
