"""Where the host time of a training step goes at the reference YAML's real shape (exp1_x4.yaml: batch 16, 32x32 crops,
T=7): wall time per step against cProfile's view of the launch path.  python tools/host_profile.py [batch] [size]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

from bench import make_net, synthetic_batch  # noqa: E402
from hipvsr.step_tail import FlatAdam  # noqa: E402
from src.runner.trainers import AcdcVSRRefineNetTrainer  # noqa: E402


def main():
    n, size = (int(a) for a in (sys.argv[1:3] + ['16', '32'][len(sys.argv) - 1:]))
    dev = torch.device('cuda:0')
    net = make_net(dev)
    tr = object.__new__(AcdcVSRRefineNetTrainer)
    tr.net, tr.loss_fns, tr.metric_fns = net, [torch.nn.L1Loss()], []
    tr.optimizer = FlatAdam(net.parameters(), lr=1e-4)
    tr.loss_weights = torch.tensor([1.0], device=dev)
    inputs, targets, pos = synthetic_batch(dev, n, 7, size, size, seed=1)
    for _ in range(3):
        tr.train_step(inputs, targets, pos)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tr.train_step(inputs, targets, pos)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'N={n} {size}x{size}: {(t2 - t0) / 10 * 1e3:.1f} ms per step wall, host enqueue {(t1 - t0) / 10 * 1e3:.1f} ms per step, '
          f'{n * 7 * 10 / (t2 - t0):.0f} frames/s')
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        tr.train_step(inputs, targets, pos)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('tottime').print_stats(18)


if __name__ == '__main__':
    main()
