set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_i
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_bf16_path.py -x -q -m gpu -k "tail or golden or inconv or in_block" > $O/tests_a.log 2>&1 || { tail -40 $O/tests_a.log; exit 1; }
tail -2 $O/tests_a.log
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench.json
python -c "
import json; d=json.load(open('$O/bench.json')); print('f32 step', d['ms_per_step'], 'ms', d['value'], 'frames/s; bf16', d['secondary']['ms_per_step'], d['secondary']['value'])"
