cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_q
mkdir -p $O
python tools/kbench_bf16.py 2>&1 | grep -v amdgpu > $O/kbench_bf16.txt
cat $O/kbench_bf16.txt
bash tools/prof_pmc_bf16.sh r03q "" > $O/pmc_bf16.log 2>&1
cp gpurun_out/pmc_r03q/summary.json $O/bf16_pmc.json
python3 - <<'P'
import json
d=json.load(open('gpurun_out/r03_q/bf16_pmc.json'))
for k,v in d.items():
    if any(x in k for x in ('conv_bf16d','wgrad_bf16','uptail','gates_bwd')):
        print(k, {c:v.get(c) for c in ('mfma_pipe_busy_frac','valu_insts_per_mfma','kernel_cycles','hbm_bytes_per_launch','SQ_LDS_BANK_CONFLICT','SQ_LDS_IDX_ACTIVE','SQ_WAIT_ANY','SQ_WAVE_CYCLES')})
P
