cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; rm -rf gpurun_out/fab; mkdir -p gpurun_out/fab
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/fab -- python tools/debug/wino_epilogue_ab.py > gpurun_out/fab/out.txt 2> gpurun_out/fab/err.txt
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/fab/*/*counter_collection.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'winoh_kernel<2>' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE']
rows.sort(key=lambda r:int(r['Dispatch_Id']))
vals=[float(r['Counter_Value']) for r in rows]
n=len(vals)//4
for i,name in enumerate(['gates+cprev','gates only','cprev only','neither']):
    seg=vals[i*n:(i+1)*n]
    print(name, 'FETCH_SIZE KB raw avg', sum(seg)/len(seg), 'x2 MB', 2*sum(seg)/len(seg)/1024)
PY
rm -rf gpurun_out/fab/*/
