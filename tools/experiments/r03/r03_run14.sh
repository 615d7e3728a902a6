set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_m
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_bf16_path.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python -m pytest tests/test_parity_r03.py -x -q -m gpu -s -k "psnr" > $O/tests_psnr.log 2>&1 || { tail -40 $O/tests_psnr.log; exit 1; }
grep "PSNR" $O/tests_psnr.log
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16.json
python -c "
import json; d=json.load(open('$O/bench_bf16.json')); print('bf16 step', d['ms_per_step'], 'ms', d['value'], 'frames/s')"
RNH_XCOL_M=0 timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16_noxcol.json
python -c "
import json; d=json.load(open('$O/bench_bf16_noxcol.json')); print('bf16 step without the frame-wise column', d['ms_per_step'], 'ms')"
