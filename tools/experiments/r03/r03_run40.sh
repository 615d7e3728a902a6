cd $GRAFT_REPO_ROOT
for f in lstm.nogates_fwd lstm.fwd lstm.nogates_fwd lstm.fwd lstm.dgrad; do python tools/kbench_bf16.py $f 2>&1 | grep -v amdgpu.ids; done
