cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ak
mkdir -p $O
python tools/wino_stamps.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -k "winograd or lstm or cfg2 or geometry or bitwise" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.txt
python tools/kbench.py lstm 2>&1 | grep -v amdgpu.ids
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/f32.json 2> $O/err.log; python -c "
import json
d=json.loads(open('$O/f32.json').read().strip().splitlines()[-1]); print('f32', d['ms_per_step'])
"
