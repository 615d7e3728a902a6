# Round 3, GPU call 2: bf16 tail kernels - their tests, the whole bf16 test file, bf16 bench line + kernel stats.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_b
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_bf16_path.py -x -q -m gpu -k "tail" > $O/tests_tail.log 2>&1 || { tail -60 $O/tests_tail.log; exit 1; }
tail -3 $O/tests_tail.log
timeout -k 10 900 python -m pytest tests/test_bf16_path.py tests/test_parity_r03.py -x -q -m gpu -s -k "not tail and not full_size" > $O/tests_bf16.log 2>&1 || { tail -60 $O/tests_bf16.log; exit 1; }
tail -3 $O/tests_bf16.log
mkdir -p $O/trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --dtype bf16 > $O/bench_bf16_profiled.json 2> $O/err.log
cp $(find $O/trace -name '*kernel_stats.csv' | head -1) $O/bf16_kernel_stats.csv
f=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python tools/step_window.py $f > $O/bf16_step_window.txt
rm -rf $O/trace
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_err.log
tail -c 1200 $O/bench_bf16.json
python profiles/summarize.py $O/bf16_kernel_stats.csv 4 | head -24
