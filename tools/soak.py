#!/usr/bin/env python3
"""Determinism soak: K training steps (forward, discounted L1, backward, FlatAdam) of the config-2 net, twice from the same seed, in
both precisions - every parameter must come out bit-identical between the two runs (no atomics, fixed-order reductions, no race), the
loss finite and decreasing on the fixed batch.   python tools/soak.py [steps_f32=12] [steps_bf16=40] [N=4]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                    # noqa: E402  (puts the package on sys.path)
import torch                                                    # noqa: E402
from hipvsr.step_tail import FlatAdam                           # noqa: E402
from src.runner.trainers import AcdcVSRRefineNetTrainer         # noqa: E402

k32 = int(sys.argv[1]) if len(sys.argv) > 1 else 12
k16 = int(sys.argv[2]) if len(sys.argv) > 2 else 40
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device('cuda:0')


def run(dtype, steps):
    net = bench.make_net(dev, seed=0)
    net.set_compute_dtype(dtype)
    opt = FlatAdam(net.parameters(), lr=1e-4, weight_decay=0)
    tr = object.__new__(AcdcVSRRefineNetTrainer)
    tr.net, tr.loss_fns, tr.metric_fns, tr.optimizer = net, [torch.nn.L1Loss()], [], opt
    tr.loss_weights = torch.tensor([1.0], device=dev)
    tr.graph, tr._graphed = False, None
    inputs, targets, pos = bench.synthetic_batch(dev, n, 7, 128, 128, seed=7)
    losses = []
    for _ in range(steps):
        _, loss, _ = tr.train_step(inputs, targets, pos)
        losses.append(float(loss))
    torch.cuda.synchronize()
    return losses, torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()


bad = 0
for dtype, steps in (('f32', k32), ('bf16', k16)):
    la, pa = run(dtype, steps)
    lb, pb = run(dtype, steps)
    same = torch.equal(pa, pb) and la == lb
    finite = all(map(lambda v: v == v and abs(v) < 1e6, la))
    print(f'{dtype}: {steps} steps x 2 runs, N = {n}: parameters bit-identical: {same}; loss {la[0]:.6f} -> {la[-1]:.6f} '
          f'({"decreasing" if la[-1] < la[0] else "NOT decreasing"}), finite: {finite}', flush=True)
    bad += (not same) + (not finite) + (la[-1] >= la[0])
sys.exit(1 if bad else 0)
