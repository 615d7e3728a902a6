cd $GRAFT_REPO_ROOT
for l in librefinenet_hip.so lib_nox.so librefinenet_hip.so lib_nox.so; do echo "== $l"; STAMPS_LIB=$l python tools/kbench.py wgrad 2>&1 | grep "wino\|lstm.wgrad\|up2"; done
