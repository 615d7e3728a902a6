cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/trc
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trc -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/trc/line.json 2> gpurun_out/trc/err.log
python - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/trc/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 40% of the trace = the timed step
n=len(rows); cut=int(n*0.55)
out=[]
for i,r in enumerate(rows):
    if i<cut: continue
    if 'copyBuffer' in r['Kernel_Name'] or 'at::native' in r['Kernel_Name'] or 'fillBuffer' in r['Kernel_Name']:
        prev=rows[i-1]['Kernel_Name'][:50]; nxt=rows[i+1]['Kernel_Name'][:50] if i+1<n else ''
        out.append((r['Kernel_Name'][:60], r.get('Grid_Size_X',r.get('Grid_Size','')), int(r['End_Timestamp'])-int(r['Start_Timestamp']), prev, nxt))
c=collections.Counter((o[0],o[1],o[3],o[4]) for o in out)
for k,v in c.most_common(40): print(v,k)
print(len(out),'non-HIP-kernel launches in the last 45% of the trace; total kernels',n)
PY
rm -f gpurun_out/trc/*/*kernel_trace.csv
