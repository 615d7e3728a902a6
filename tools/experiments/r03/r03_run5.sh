set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_e
mkdir -p $O
PK=$GRAFT_REPO_ROOT/efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd
for rep in 1 2; do for v in 0 1 2; do echo "== RNH_SCHED=$v"; RNH_LIB=$PK/hipvsr/lib_s$v.so python tools/kbench_bf16.py 2>&1 | grep -v "amdgpu.ids\|wgrad\|gates_bwd"; done; done > $O/sched.txt
cat $O/sched.txt
timeout -k 10 600 python -m pytest tests/test_bf16_path.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16.json
tail -c 600 $O/bench_bf16.json
