"""Timing experiment (results are garbage, only the clock counts): the step of bench.py with one engine operation turned into a
no-op - an upper bound on what removing / fusing that launch could save.  The first REAL calls run the real operation (give the
number of calls of the warm-up steps), so that the buffers it no longer writes hold values of realistic magnitude - matrix-core
power, and with it the clock, depends on the data.

  python tools/exp_without.py lstm_gates_bwd 378 -- --steps 10 --warmup 3 --no-cpu-baseline [--dtype bf16 --no-secondary]
"""
import sys

sys.path.insert(0, __file__.rsplit('/', 2)[0])
import bench                                                    # noqa: E402  (puts the package on sys.path)
from hipvsr import hip_ops                                      # noqa: E402

cut = sys.argv.index('--')
name, real, rest = sys.argv[1], int(sys.argv[2]), sys.argv[cut + 1:]
orig = getattr(hip_ops.HipOps, name)
calls = [0]


def maybe(self, *a, **k):
    calls[0] += 1
    if calls[0] % 100000 <= real:                              # (the secondary case starts a new count)
        return orig(self, *a, **k)
    return None


setattr(hip_ops.HipOps, name, maybe)
sys.argv = ['bench.py'] + rest
print('without', name, 'after', real, 'calls', file=sys.stderr)
bench.main()
print('calls', calls[0], file=sys.stderr)
