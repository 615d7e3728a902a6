#!/usr/bin/env python3
"""Training step at the REFERENCE's own training shape (configs/train/refine_net/exp1_x4.yaml:21-33: batch 16, crops of
32 x 32 -> 128 x 128, T = 7, F = 19, exp1_x4 net): eager launches against the HIP-graph replay of forward + loss + backward
(hipvsr.graph.GraphedTrainStep), fp32 and bf16 storage.  GPU box only.   python tools/train_shape_bench.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd')
for p in (ROOT, PKG):
    sys.path.insert(0, p)
import torch                                                    # noqa: E402
from bench import make_net, synthetic_batch                     # noqa: E402
from hipvsr.step_tail import FlatAdam                           # noqa: E402
from src.runner.trainers import AcdcVSRRefineNetTrainer         # noqa: E402

dev = torch.device('cuda:0')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N, T, H = 16, 7, 32
inputs, targets, pos = synthetic_batch(dev, N, T, H, H, seed=3)
for dtype in ('f32', 'bf16'):
    for graph in (False, True):
        net = make_net(dev, seed=0).set_compute_dtype(dtype)
        tr = object.__new__(AcdcVSRRefineNetTrainer)
        tr.net, tr.loss_fns, tr.metric_fns, tr.graph, tr._graphed = net, [torch.nn.L1Loss()], [], graph, None
        tr.loss_weights = torch.tensor([1.0], device=dev)
        tr.optimizer = FlatAdam(net.parameters(), lr=1e-4)
        for _ in range(3):
            _, loss, _ = tr.train_step(inputs, targets, pos)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            _, loss, _ = tr.train_step(inputs, targets, pos)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        print(f'{dtype:5s} {"graph" if graph else "eager":6s} {ms:8.2f} ms/step  {N * T / ms * 1e3:8.1f} supervised frames/s   loss {float(loss):.6f}', flush=True)
