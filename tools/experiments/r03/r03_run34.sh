cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ae
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt
python bench.py > $O/bench_line.json 2> $O/bench.err && python -c "
import json
d=json.loads(open('$O/bench_line.json').read().strip().splitlines()[-1]); print('f32', d['ms_per_step'], d['value'], 'bf16', d['secondary']['ms_per_step'], d['secondary']['value'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], d['secondary']['roofline']['traffic'])
"
