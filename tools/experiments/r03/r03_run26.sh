cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_y
mkdir -p $O
python tools/kbench_bf16.py lstm > $O/kbench_bf16_lstm.txt 2>&1
bash tools/prof_pmc_wino.sh r03y > $O/pmc_wino.log 2>&1
bash tools/prof_pmc_bf16.sh r03ybf lstm > $O/pmc_bf16.log 2>&1
cp gpurun_out/pmc_r03y/summary.json $O/wino_pmc.json
cp gpurun_out/pmc_r03y/lstm_kernel_hbm_bytes.json $O/lstm_kernel_hbm_bytes.json
cp gpurun_out/pmc_r03ybf/summary.json $O/bf16_pmc.json
grep -v amdgpu.ids $O/kbench_bf16_lstm.txt
cat $O/lstm_kernel_hbm_bytes.json
