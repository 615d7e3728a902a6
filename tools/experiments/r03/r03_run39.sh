cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ai
mkdir -p $O
for m in layer cell dir layer cell; do
  RNH_LSTM_STREAMS=$m python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/f32_$m.json 2> $O/err.log
  python -c "
import json
d=json.loads(open('$O/f32_$m.json').read().strip().splitlines()[-1]); print('$m', d['ms_per_step'])
"
done
