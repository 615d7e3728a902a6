set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_j
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_bf16_path.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for rep in 1 2; do
echo "== KC=32 where possible"; timeout -k 10 120 python tools/kbench_bf16.py 2>&1 | grep -v "amdgpu.ids\|wgrad\|gates_bwd"
echo "== KC=16 everywhere";     RNH_BF16_KC=1 timeout -k 10 120 python tools/kbench_bf16.py 2>&1 | grep -v "amdgpu.ids\|wgrad\|gates_bwd"
done | tee $O/kbench_kc.txt
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16.json
python -c "
import json; d=json.load(open('$O/bench_bf16.json')); print('bf16 step', d['ms_per_step'], 'ms', d['value'], 'frames/s', d['roofline']['avg_launch_ms'])"
