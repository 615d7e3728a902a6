cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_aj
mkdir -p $O
python tools/soak.py 12 40 4 > $O/soak.txt 2>&1; echo "soak rc=$?"; grep -v amdgpu.ids $O/soak.txt | tail -4
