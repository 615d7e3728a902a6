cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_w
mkdir -p $O
for s in 1 2 3 6; do python tools/kbench_streams.py $s 20; done > $O/streams.txt 2>&1
for sk in 20 40 60; do RNH_SKEW_BWD_US=$sk python tools/kbench_streams.py 3 20; done >> $O/streams.txt 2>&1
cat $O/streams.txt | grep -v amdgpu.ids
