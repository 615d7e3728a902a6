cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_z
mkdir -p $O
which rocm-smi amd-smi > $O/which.txt 2>&1
rocm-smi --showclocks --showpower --showtemp > $O/smi_idle.txt 2>&1
for dt in bf16 f32; do
  ( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | tr '\n' ' '; echo; sleep 0.5; done ) > $O/smi_$dt.txt &
  SMI=$!
  steps=150; [ $dt = f32 ] && steps=45
  python bench.py --steps $steps --warmup 3 --no-cpu-baseline --no-secondary --dtype $dt > $O/bench_$dt.json 2> $O/bench_$dt.err
  wait $SMI
done
head -30 $O/smi_idle.txt
sed -n 8,20p $O/smi_bf16.txt
sed -n 8,20p $O/smi_f32.txt
