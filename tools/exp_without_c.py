"""Timing experiment (results are garbage, only the clock counts): the step of bench.py with ONE C-ABI entry point of librefinenet_hip.so turned
into a no-op after its first <real> calls (the warm-up steps run the real thing, so buffers keep realistic values) - an upper bound on what
that launch costs the step.   python tools/exp_without_c.py rnh_wgrad_reduce 90 -- --steps 10 --warmup 3 --no-cpu-baseline --dtype bf16"""
import sys

sys.path.insert(0, __file__.rsplit('/', 2)[0])
import bench                                                    # noqa: E402  (puts the package on sys.path)
from hipvsr import lib as L                                     # noqa: E402

cut = sys.argv.index('--')
name, real, rest = sys.argv[1], int(sys.argv[2]), sys.argv[cut + 1:]
lib = L.load()
orig = getattr(lib, name)
calls = [0]


def maybe(*a):
    calls[0] += 1
    return orig(*a) if calls[0] <= real else 0


setattr(lib, name, maybe)
sys.argv = ['bench.py'] + rest
bench.main()
print('calls of', name, calls[0], file=sys.stderr)
