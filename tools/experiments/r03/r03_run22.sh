cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_u
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_bf16_path.py -x -q -k "fused" > $O/pytest_fused.txt 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest_fused.txt
run() {
  name=$1; shift; dt=$1; shift
  env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --dtype $dt > $O/$name.json 2> $O/$name.err
  python -c "
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['ms_per_step'])
except Exception as e: print('$name failed', e)
"
}
run bf16_fused bf16 A=1 &&
run bf16_unfused bf16 RNH_FUSE_GATES_BWD=0 &&
run bf16_fused_cell bf16 RNH_LSTM_STREAMS=cell
