cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_r
mkdir -p $O
python tools/exp_without.py lstm_gates_bwd 378 -- --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/no_gates_bwd_f32.json 2> $O/no_gates_bwd_f32.err &&
python tools/exp_without.py lstm_gates_bwd 378 -- --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --dtype bf16 > $O/no_gates_bwd_bf16.json 2> $O/no_gates_bwd_bf16.err &&
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --dtype bf16 > $O/base_bf16.json 2> $O/base_bf16.err
tail -2 $O/*.err
python - <<'PY'
import json
for f in ('no_gates_bwd_f32', 'no_gates_bwd_bf16', 'base_bf16'):
    try:
        d = json.loads(open('gpurun_out/r03_r/%s.json' % f).read().strip().splitlines()[-1])
        print(f, 'ms', d['ms_per_step'])
    except Exception as e:
        print(f, 'failed', e)
PY
