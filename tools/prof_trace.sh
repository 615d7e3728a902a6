set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_t
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_t -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_t/bench_line.json 2> gpurun_out/prof_t/err.log
f=$(find gpurun_out/prof_t -name '*kernel_trace.csv' | head -1)
# the timed steps are the middle of the trace: skip the warm-up quarter, the roofline launches at the end are short
python tools/trace_attrib.py $f 0.3 > gpurun_out/prof_t/attrib.txt
rm -f $f
cat gpurun_out/prof_t/attrib.txt
