cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ao
mkdir -p $O
for l in librefinenet_hip.so lib_nt.so librefinenet_hip.so lib_nt.so; do
  python tools/bench_with_lib.py $l -- --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/f32_$l.json 2> $O/err.log
  python -c "
import json
d=json.loads(open('$O/f32_$l.json').read().strip().splitlines()[-1]); print('$l', d['ms_per_step'])
"
done
