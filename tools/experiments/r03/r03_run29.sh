cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/prof_pmc_bf16.sh r03zbf lstm.fwd > gpurun_out/pmc_r03zbf.log 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_r03zbf/summary.json'))
for k,v in d.items():
    if 'conv_bf16d' in k: print(k, json.dumps(v))
PY
