set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_j2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_j2 -- python tools/kbench.py lstm.fwd > gpurun_out/prof_j2/kbench.log 2> gpurun_out/prof_j2/err.log
find gpurun_out/prof_j2 -name '*kernel_trace.csv' -delete
cat gpurun_out/prof_j2/kbench.log | grep -v amdgpu
f=$(find gpurun_out/prof_j2 -name '*kernel_stats.csv' | head -1); head -5 $f | cut -c1-200
