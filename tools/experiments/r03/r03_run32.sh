cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/prof_pmc_wino.sh r03ac > gpurun_out/pmc_r03ac.log 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_r03ac/summary.json'))
for k,v in d.items():
    if 'wgrad' in k or 'winoh' in k: print(k[:50], json.dumps(v))
PY
