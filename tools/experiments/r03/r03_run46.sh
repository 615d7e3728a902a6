cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_am
mkdir -p $O
RNH_FUSE_GATES_BWD=0 RNH_WGRAD_HALF=0 timeout -k 10 1100 python -m pytest tests -q -m gpu > $O/pytest_gpu_fallbacks.txt 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest_gpu_fallbacks.txt
