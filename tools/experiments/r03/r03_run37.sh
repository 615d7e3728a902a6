cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ag
mkdir -p $O
python bench.py > $O/bench_line.json 2> $O/bench.err && python -c "
import json
d=json.loads(open('$O/bench_line.json').read().strip().splitlines()[-1]); print('f32', d['ms_per_step'], d['value'], 'bf16', d['secondary']['ms_per_step'], d['secondary']['value'], 'frac', d['roofline']['frac'], d['roofline']['avg_launch_ms'], 'cfg', d['config']['executed_frac_of_f32_mfma_peak'], d['secondary']['config']['algorithmic_frac_of_bf16_mfma_peak'], 'cpu', d['cpu_baseline']['value'])
"
