cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_k
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/nt_store_probe.hip -o /tmp/nt_probe 2>/dev/null && timeout -k 10 300 /tmp/nt_probe > $O/nt_store_probe.txt 2>&1
cat $O/nt_store_probe.txt
bash tools/prof_lstm_isolated.sh > $O/lstm_isolated.log 2>&1
cp $(find gpurun_out/prof_l2 -name '*kernel_stats.csv' | head -1) $O/lstm_kernel_isolated_stats.csv
tail -8 $O/lstm_isolated.log
bash tools/prof_pmc_wino.sh r03k > $O/pmc_wino.log 2>&1
cp gpurun_out/pmc_r03k/summary.json $O/wino_pmc.json; cp gpurun_out/pmc_r03k/lstm_kernel_hbm_bytes.json $O/lstm_kernel_hbm_bytes.json
python3 - <<'P'
import json
d=json.load(open('gpurun_out/r03_k/wino_pmc.json'))
for k,v in d.items():
    if 'wino' in k and 'pack' not in k and 'sum' not in k and 'reduce' not in k:
        print(k, {c:v.get(c) for c in ('mfma_pipe_busy_frac','valu_insts_per_mfma','kernel_cycles','hbm_bytes_per_launch')})
P
