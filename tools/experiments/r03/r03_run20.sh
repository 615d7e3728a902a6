cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_s
mkdir -p $O
run() {  # name, env..., dtype
  name=$1; shift; dt=$1; shift
  env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --dtype $dt > $O/$name.json 2> $O/$name.err
  python -c "
import json,sys
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['ms_per_step'])
except Exception as e: print('$name failed', e)
"
}
run bf16_base bf16 A=1 &&
run bf16_cell bf16 RNH_LSTM_STREAMS=cell &&
run bf16_split2 bf16 RNH_BPTT_SPLIT=2 &&
run bf16_split2_cell bf16 RNH_BPTT_SPLIT=2 RNH_LSTM_STREAMS=cell &&
run bf16_split4 bf16 RNH_BPTT_SPLIT=4 &&
run f32_split2 f32 RNH_BPTT_SPLIT=2
