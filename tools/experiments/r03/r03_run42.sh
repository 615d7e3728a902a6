cd $GRAFT_REPO_ROOT
for i in 1 2; do python tools/wino_stamps.py 2>&1 | grep -v amdgpu.ids; echo; done
NOGATES=1 python tools/wino_stamps.py 2>&1 | grep -v amdgpu.ids
