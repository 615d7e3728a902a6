cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_o
mkdir -p $O
python tools/kbench_bf16.py > $O/kbench_bf16.txt 2>&1
bash tools/prof_pmc_bf16.sh r03o "" > $O/pmc_bf16.log 2>&1
cp gpurun_out/pmc_r03o/summary.json $O/bf16_pmc.json
for dt in f32 bf16; do
  mkdir -p $O/trace_$dt
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$dt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --dtype $dt > $O/trace_$dt/bench_line.json 2> $O/trace_$dt/err.log
  f=$(find $O/trace_$dt -name '*kernel_trace.csv' | head -1)
  python tools/step_window.py $f > $O/${dt}_step_window.txt
  cp $(find $O/trace_$dt -name '*kernel_stats.csv' | head -1) $O/${dt}_kernel_stats.csv
  rm -rf $O/trace_$dt
done
python profiles/summarize.py $O/f32_kernel_stats.csv 4 | head -40
python profiles/summarize.py $O/bf16_kernel_stats.csv 4 | head -32
