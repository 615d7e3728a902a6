cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_x
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest_gpu.txt
