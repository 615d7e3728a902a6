cd $GRAFT_REPO_ROOT
HWMAP=1 python tools/wino_stamps.py 2>&1 | grep -v amdgpu.ids
