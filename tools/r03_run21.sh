cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_t
mkdir -p $O/trace
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --dtype bf16 > $O/bench_line.json 2> $O/err.log
f=$(find $O/trace -name '*kernel_trace.csv' | head -1)
head -2 $f > $O/trace_head.txt
python tools/step_timeline.py $f > $O/bf16_timeline.txt
rm -rf $O/trace
wc -l $O/bf16_timeline.txt
