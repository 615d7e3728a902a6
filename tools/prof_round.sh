# The round's profile set in one GPU-box call:  bash tools/prof_round.sh <tag>      -> gpurun_out/<tag>/
#   bench lines (default = config 2 + bf16 secondary + cpu baseline; config 4; config 5; the reference YAML's shape eager and graph-replayed), rocprofv3
#   kernel stats + one-step windows of the fp32 and the bf16 step, the isolated ConvLSTM cell launches (fp32 Winograd, bf16) under --kernel-trace --stats,
#   the per-kernel micro-benchmarks.  (PMC passes: tools/prof_pmc_wino.sh, tools/prof_pmc_bf16.sh - counters never share a run with traces.)
set +e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/${1:-round}
mkdir -p $o
python bench.py > $o/bench_line.json 2> $o/bench_line.err
python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline > $o/cfg4_bench_line.json 2> $o/cfg4.err
python bench.py --config 5 --no-cpu-baseline > $o/cfg5_bench_line.json 2> $o/cfg5.err
python bench.py --config yaml --no-cpu-baseline --steps 20 --warmup 5 > $o/yaml_bench_line_eager.json 2> $o/yaml_eager.err
python bench.py --config yaml --no-cpu-baseline --steps 20 --warmup 5 --graph on > $o/yaml_bench_line_graph.json 2> $o/yaml_graph.err
for dt in f32 bf16; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $o/step_$dt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --dtype $dt > $o/step_$dt.json 2> $o/step_$dt.err
    f=$(find $o/step_$dt -name '*kernel_trace.csv' | head -1)
    python tools/step_window.py $f > $o/${dt}_step_window.txt 2>&1
    cp $(find $o/step_$dt -name '*kernel_stats.csv' | head -1) $o/${dt}_kernel_stats.csv
    rm -rf $o/step_$dt
done
rocprofv3 --kernel-trace --stats --output-format csv -d $o/iso_f32 -- python tools/lstm_roofline.py > $o/lstm_kernel_isolated_line.json 2> $o/iso_f32.err
cp $(find $o/iso_f32 -name '*kernel_stats.csv' | head -1) $o/lstm_kernel_isolated_stats.csv; rm -rf $o/iso_f32
rocprofv3 --kernel-trace --stats --output-format csv -d $o/iso_bf16 -- python tools/kbench_bf16.py lstm.fwd > $o/bf16_cell_isolated.txt 2> $o/iso_bf16.err
cp $(find $o/iso_bf16 -name '*kernel_stats.csv' | head -1) $o/bf16_cell_isolated_stats.csv; rm -rf $o/iso_bf16
python tools/kbench.py > $o/kbench_f32.txt 2>&1
python tools/kbench_bf16.py > $o/kbench_bf16.txt 2>&1
ls -la $o
