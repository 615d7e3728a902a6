# Round 3, GPU call 1: the new parity tests, the bench line with its bf16 secondary, step anatomy of the fp32 and the bf16 step.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_a
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_r03.py -x -q -m gpu -s -k "full_size or lstm_cell or gates_bwd" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -5 $O/tests.log
timeout -k 10 400 python bench.py > $O/bench_line.json 2> $O/bench_err.log
tail -c 3000 $O/bench_line.json
for dt in f32 bf16; do
  mkdir -p $O/trace_$dt
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$dt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --dtype $dt > $O/trace_$dt/bench_line.json 2> $O/trace_$dt/err.log
  f=$(find $O/trace_$dt -name '*kernel_trace.csv' | head -1)
  python tools/step_window.py $f > $O/${dt}_step_window.txt
  python tools/trace_attrib.py $f 0.3 > $O/${dt}_attrib.txt
  cp $(find $O/trace_$dt -name '*kernel_stats.csv' | head -1) $O/${dt}_kernel_stats.csv
  gzip -c $f > $O/${dt}_kernel_trace.csv.gz
  rm -rf $O/trace_$dt
  head -5 $O/${dt}_step_window.txt
done
