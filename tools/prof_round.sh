# Round profile: rocprofv3 kernel stats of the bench command (-> profiles/r01_<x>_bench_kernel_stats.csv) and the full bench line.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_o
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_o -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_o/bench_line_profiled.json 2> gpurun_out/prof_o/err.log
find gpurun_out/prof_o -name '*kernel_trace.csv' -delete
python bench.py > gpurun_out/prof_o/bench_line.json 2> gpurun_out/prof_o/bench_err.log
tail -c 1500 gpurun_out/prof_o/bench_line.json
