# One-step windows of the fp32 and the bf16 training step under rocprofv3 --kernel-trace --stats (GPU box):  [BENCH_ARGS="--config yaml"] bash tools/prof_step.sh <out dir under gpurun_out> [f32|bf16 ...]
set +e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/$1; shift
mkdir -p $o
for dt in ${@:-f32 bf16}; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $o/step_$dt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --dtype $dt $BENCH_ARGS > $o/step_$dt.json 2> $o/step_$dt.err
    f=$(find $o/step_$dt -name '*kernel_trace.csv' | head -1)
    python tools/step_window.py $f > $o/${dt}_step_window.txt 2>&1
    python tools/grid_rounds.py $f > $o/${dt}_grid_rounds.txt 2>&1
    cp $(find $o/step_$dt -name '*kernel_stats.csv' | head -1) $o/${dt}_kernel_stats.csv
    rm -rf $o/step_$dt
done
