set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_f
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_bf16_path.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python tools/kbench_bf16.py wgrad 2>&1 | grep -v amdgpu.ids | tee $O/kbench_wgrad.txt
python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16.json
python -c "
import json; d=json.load(open('$O/bench_bf16.json')); print('bf16 step', d['ms_per_step'], 'ms', d['value'], 'frames/s')"
