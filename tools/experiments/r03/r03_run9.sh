set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_h
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_bf16_path.py -x -q -m gpu -k "wgrad or weight" > $O/tests_w.log 2>&1 || { tail -40 $O/tests_w.log; exit 1; }
tail -2 $O/tests_w.log
timeout -k 10 120 python tools/kbench_bf16.py wgrad 2>&1 | grep -v amdgpu.ids | tee $O/kbench_wgrad.txt
RNH_WGRAD_DMA=0 timeout -k 10 120 python tools/kbench_bf16.py wgrad 2>&1 | grep -v amdgpu.ids | tee -a $O/kbench_wgrad.txt
timeout -k 10 600 python -m pytest tests/test_bf16_path.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16.json
python -c "
import json; d=json.load(open('$O/bench_bf16.json')); print('bf16 step', d['ms_per_step'], 'ms', d['value'], 'frames/s')"
