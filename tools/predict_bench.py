"""Whole-cycle inference (row f2) at the reference's test shape: batch 1, one cardiac cycle of 30 frames (F = 42 input
frames), 54 x 64 -> 216 x 256, exp1_x4 net.  Times the forward as the reference's predictor runs it (all 9 output
groups, eager), with only the consumed group (last_group_only) and replayed from a HIP graph.
  python tools/predict_bench.py [cycle_frames] [H] [W]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

from bench import make_net, synthetic_batch  # noqa: E402
from hipvsr.graph import GraphedForward  # noqa: E402


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    Tc, H, W = (int(a) for a in (sys.argv[1:4] + ['30', '54', '64'][len(sys.argv) - 1:]))
    dev = torch.device('cuda:0')
    net = make_net(dev).eval()
    inputs, _, pos = synthetic_batch(dev, 1, Tc, H, W, seed=1)
    with torch.no_grad():
        net.last_group_only = False
        a = timed(lambda: net(inputs, pos))
        net.last_group_only = True
        b = timed(lambda: net(inputs, pos))
        gf = GraphedForward(net)
        c = timed(lambda: gf(inputs, pos))
    for name, ms in (('eager, all 9 groups (the reference predictor\'s call)', a), ('eager, last group only', b), ('HIP graph, last group only', c)):
        print(f'{name}: {ms:.2f} ms per cycle = {Tc / ms * 1e3:.0f} frames/s')
    for K in (2, 4, 8, 16):                               # K cines of one shape as one batch (the predictor's cines_per_launch)
        inputs, _, pos = synthetic_batch(dev, K, Tc, H, W, seed=1)
        with torch.no_grad():
            d = timed(lambda: gf(inputs, pos))
        print(f'HIP graph, last group only, {K} cines per launch: {d:.2f} ms = {K * Tc / d * 1e3:.0f} frames/s')


if __name__ == '__main__':
    main()
