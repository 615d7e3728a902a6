cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_final
mkdir -p $O
python bench.py > $O/bench_line.json 2> $O/bench.err
tail -1 $O/bench_line.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
