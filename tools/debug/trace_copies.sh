# Which kernels of a steady-state training step are NOT this library's (ATen element-wise kernels, runtime copies / fills)?
#   usage (GPU box): bash tools/debug/trace_copies.sh [bench args]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/trc; rm -rf gpurun_out/trc/*
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trc -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/trc/line.json 2> gpurun_out/trc/err.log
python - <<'PY'
import csv,glob,collections,re
f=glob.glob('gpurun_out/trc/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last adam_kernel launches delimit steps: take the kernels between the 2nd-last and the last optimizer step
ad=[i for i,r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
lo,hi=ad[-3]+1,ad[-1]+1          # two adam launches per step
step=rows[lo:hi]
mine=lambda n: '(anonymous namespace)' in n and 'at::native' not in n
c=collections.Counter()
for r in step:
    n=r['Kernel_Name']
    if mine(n): continue
    short=re.sub(r'\(.*','',n); short=re.sub(r'<.*','',short)
    m=re.search(r'(FillFunctor|MulFunctor|CUDAFunctor_add|MeanOps|sum_functor|CatArray|direct_copy|copyBuffer|fillBuffer|AUnaryFunctor|BUnaryFunctor|BinaryFunctor)',n)
    c[(short[-50:], m.group(1) if m else '', r.get('Grid_Size_X', r.get('Grid_Size','')))]+=1
for k,v in c.most_common(): print(v,k)
print(sum(c.values()),'foreign launches of',len(step),'in one steady-state step')
PY
rm -f gpurun_out/trc/*/*kernel_trace.csv
