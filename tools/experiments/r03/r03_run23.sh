cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_v
mkdir -p $O
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
run base bf16 A=1 &&
run bwd20 bf16 RNH_SKEW_BWD_US=20 &&
run bwd40 bf16 RNH_SKEW_BWD_US=40 &&
run bwd60 bf16 RNH_SKEW_BWD_US=60 &&
run fwd15 bf16 RNH_SKEW_FWD_US=15 &&
run fwd30 bf16 RNH_SKEW_FWD_US=30 &&
run fwd45 bf16 RNH_SKEW_FWD_US=45 &&
run both bf16 RNH_SKEW_FWD_US=30 RNH_SKEW_BWD_US=40
