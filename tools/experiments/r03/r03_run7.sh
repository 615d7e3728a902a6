cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/prof_pmc_bf16.sh r03g lstm.wgrad > gpurun_out/pmc_r03g.log 2>&1
python3 - <<'P'
import json
d=json.load(open('gpurun_out/pmc_r03g/summary.json'))
for k,v in d.items():
    if 'wgrad_bf16' in k:
        print(k)
        for c,x in v.items(): print('   ',c,x)
P
