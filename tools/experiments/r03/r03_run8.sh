cd $GRAFT_REPO_ROOT
PK=$GRAFT_REPO_ROOT/efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd
for rep in 1 2; do for m in 0 1 2 3 4 7; do L=$PK/hipvsr/lib_w$m.so; [ $m = 0 ] && L=$PK/hipvsr/librefinenet_hip.so; echo "== WEXP=$m"; RNH_LIB=$L python tools/kbench_bf16.py lstm.wgrad 2>&1 | grep -v amdgpu; done; done
