set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_p
mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_hip_parity.py tests/test_bf16_path.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench.json
python -c "
import json; d=json.load(open('$O/bench.json')); print('f32 step', d['ms_per_step'], 'ms', d['value'], 'frames/s; bf16', d['secondary']['ms_per_step'], d['secondary']['value'])"
