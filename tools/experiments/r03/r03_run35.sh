cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_af
mkdir -p $O
RNH_BF16_MB=2 timeout -k 10 600 python -m pytest tests/test_bf16_path.py -x -q -k "conv_bf16_kernel or fused or ragged" > $O/pytest_mb2.txt 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest_mb2.txt
for mb in 4 2 4 2; do echo "MB=$mb"; RNH_BF16_MB=$mb python tools/kbench_bf16.py lstm 2>&1 | grep -v amdgpu.ids; done > $O/kbench_mb.txt
cat $O/kbench_mb.txt
for mb in 4 2 4 2; do
  RNH_BF16_MB=$mb python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --dtype bf16 > $O/bf16_mb$mb.json 2> $O/err.log
  python -c "
import json
d=json.loads(open('$O/bf16_mb$mb.json').read().strip().splitlines()[-1]); print('mb=$mb', d['ms_per_step'])
"
done
