cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ab
mkdir -p $O
for h in 1 0 1 0; do
  RNH_WGRAD_HALF=$h python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/f32_half$h.json 2> $O/err.log
  python -c "
import json
d=json.loads(open('$O/f32_half$h.json').read().strip().splitlines()[-1]); print('half=$h', d['ms_per_step'])
"
done
