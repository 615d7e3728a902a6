set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_l2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_l2 -- python tools/lstm_roofline.py > gpurun_out/prof_l2/kbench.log 2> gpurun_out/prof_l2/err.log
find gpurun_out/prof_l2 -name '*kernel_trace.csv' -delete
cat gpurun_out/prof_l2/kbench.log | grep -v amdgpu
f=$(find gpurun_out/prof_l2 -name '*kernel_stats.csv' | head -1); head -5 $f | cut -c1-200
