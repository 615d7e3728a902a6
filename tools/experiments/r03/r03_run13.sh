cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_l
mkdir -p $O
timeout -k 10 400 python tools/debug/nt_gates_check.py lib_nt.so 2>&1 | grep -v amdgpu | tail -3 | tee $O/nt_gates.txt
for rep in 1 2 3; do for l in librefinenet_hip.so lib_nt.so; do echo "== $l"; STAMPS_LIB=$l timeout -k 10 120 python tools/kbench.py lstm 2>&1 | grep -v amdgpu | grep -i "lstm" | head -4; done; done | tee -a $O/nt_gates.txt
