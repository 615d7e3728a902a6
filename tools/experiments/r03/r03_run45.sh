cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_al
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_parity_r03.py -x -q -k "winograd or lstm_cell or cfg2_geometry or bench_launch" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
for p in 1 0 1 0; do echo "persist=$p"; RNH_WINO_PERSIST=$p timeout -k 10 120 python tools/kbench.py lstm.fwd 2>&1 | grep -v amdgpu.ids; done
for p in 1 0; do
  RNH_WINO_PERSIST=$p timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/f32_p$p.json 2> $O/err.log; python -c "
import json
d=json.loads(open('$O/f32_p$p.json').read().strip().splitlines()[-1]); print('persist=$p', d['ms_per_step'])
"
done
