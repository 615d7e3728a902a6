cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_aa
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -k "weight_gradient" > $O/pytest_wgrad.txt 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest_wgrad.txt
for h in 1 0 1 0; do echo "RNH_WGRAD_HALF=$h"; RNH_WGRAD_HALF=$h python tools/kbench.py wgrad 2>&1 | grep -v amdgpu.ids; done > $O/kbench_wgrad.txt
cat $O/kbench_wgrad.txt
