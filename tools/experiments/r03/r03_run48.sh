cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_an
mkdir -p $O/t
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python tools/kbench_bf16.py lstm.fwd > $O/kbench.txt 2> $O/err.log
cp $(find $O/t -name '*kernel_stats.csv' | head -1) $O/bf16_cell_isolated_stats.csv
rm -rf $O/t
grep -v amdgpu.ids $O/kbench.txt; head -4 $O/bf16_cell_isolated_stats.csv | cut -c1-200
