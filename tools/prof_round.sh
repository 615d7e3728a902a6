set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_i gpurun_out/prof_pred
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_i -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_i/bench_line.json 2> gpurun_out/prof_i/err.log
find gpurun_out/prof_i -name '*kernel_trace.csv' -delete
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_pred -- python tools/predict_bench.py > gpurun_out/prof_pred/out.log 2> gpurun_out/prof_pred/err.log
f=$(find gpurun_out/prof_pred -name '*kernel_trace.csv' | head -1)
python tools/trace_gaps.py $f 0.85 > gpurun_out/prof_pred/gaps_last15pct.txt
python - "$f" <<'P' > gpurun_out/prof_pred/overlap.txt
import csv, sys
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:50]) for r in csv.DictReader(open(sys.argv[1])))
t0, t1 = rows[0][0], rows[-1][1]
rows = [r for r in rows if r[0] > t0 + 0.9 * (t1 - t0)]
# sum of durations vs union: > 1 means concurrency
tot = sum(e - s for s, e, _ in rows)
busy, cs, ce = 0, rows[0][0], rows[0][1]
for s, e, _ in rows[1:]:
    if s > ce:
        busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f'last 10% of the trace (graph replays): {len(rows)} kernels, sum of durations {tot/1e6:.2f} ms, union {busy/1e6:.2f} ms, span {(rows[-1][1]-rows[0][0])/1e6:.2f} ms, concurrency {tot/busy:.2f}')
import collections
c = collections.Counter(); d = collections.Counter()
for s, e, n in rows: c[n] += 1; d[n] += e - s
for n, v in d.most_common(12): print(f'{n:50s} calls {c[n]:6d} avg {v/c[n]/1e3:8.1f} us total {v/1e6:8.2f} ms')
P
rm -f $f
cat gpurun_out/prof_pred/overlap.txt gpurun_out/prof_pred/gaps_last15pct.txt gpurun_out/prof_pred/out.log
tail -c 400 gpurun_out/prof_i/bench_line.json
