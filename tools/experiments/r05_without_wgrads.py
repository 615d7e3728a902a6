"""Timing experiment (results are garbage, only the clock counts): bench.py's step with the weight gradients whose plan name matches a regular expression turned into
no-ops - what those launches cost IN the step (an upper bound on what a faster form of them could save).
    python tools/experiments/r05_without_wgrads.py '<regex>' -- --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                                    # noqa: E402  (puts the package on sys.path)
from hipvsr import hip_ops                                      # noqa: E402

cut = sys.argv.index('--')
pat, rest = re.compile(sys.argv[1]), sys.argv[cut + 1:]
orig = hip_ops.HipOps.wgrad
seen = {}


def maybe(self, plan, *a, **k):
    hit = bool(pat.search(plan.name))
    seen[plan.name] = seen.get(plan.name, 0) + 1
    if hit:
        return None
    return orig(self, plan, *a, **k)


hip_ops.HipOps.wgrad = maybe
sys.argv = ['bench.py'] + rest
bench.main()
print('skipped:', sorted(n for n in seen if pat.search(n)), '| kept:', sorted(n for n in seen if not pat.search(n)), file=sys.stderr)
