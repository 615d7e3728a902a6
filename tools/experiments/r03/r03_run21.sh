cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_t
for mode in two_launches fused; do
  mkdir -p $O/trace_$mode
  if [ $mode = two_launches ]; then export RNH_FUSE_GATES_BWD=0; else export RNH_FUSE_GATES_BWD=1; fi
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$mode -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --dtype bf16 > $O/bench_line_$mode.json 2> $O/err_$mode.log
  f=$(find $O/trace_$mode -name '*kernel_trace.csv' | head -1)
  python tools/step_timeline.py $f > $O/bf16_timeline_$mode.txt
  python tools/step_window.py $f > $O/bf16_step_window_$mode.txt
  rm -rf $O/trace_$mode
done
wc -l $O/*.txt
