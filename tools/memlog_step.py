#!/usr/bin/env python3
"""One training step at a BASELINE config with RNH_MEMLOG=1: allocated HBM at the engine's stage boundaries beside
engine.memory_plan()'s estimate (calibration of the 'auto' gate-memory plan).  usage: memlog_step.py <config 2|4|5> <f32|bf16> [RNH_GATES value]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['RNH_MEMLOG'] = '1'
if len(sys.argv) > 3:
    os.environ['RNH_GATES'] = sys.argv[3]
import bench      # noqa: E402
import torch      # noqa: E402

args = bench.parse_args(['--config', sys.argv[1]])
dev = torch.device('cuda:0')
net = bench.make_net(dev, scale=args.scale).set_compute_dtype(sys.argv[2])
inputs, targets, pos = bench.synthetic_batch(dev, args.batch, args.frames, args.size, args.size, seed=1, s=args.scale)
from src.runner.trainers import AcdcVSRRefineNetTrainer      # noqa: E402
tr = object.__new__(AcdcVSRRefineNetTrainer)
tr.net, tr.loss_fns, tr.metric_fns, tr.optimizer = net, [torch.nn.L1Loss()], [], torch.optim.SGD(net.parameters(), lr=0.0)
tr.loss_weights = torch.tensor([1.0], device=dev)
tr.graph, tr._graphed = False, None
torch.cuda.reset_peak_memory_stats(dev)
base = torch.cuda.memory_allocated(dev)
tr.train_step(inputs, targets, pos)
torch.cuda.synchronize()
eng = net._engine()
n = eng.recompute_stages(args.batch, args.size, args.size, args.frames + 12)
plan = eng.memory_plan(args.batch, args.size, args.size, args.frames + 12, recompute=n)
print(f'config {sys.argv[1]} {sys.argv[2]} RNH_GATES={os.environ.get("RNH_GATES", "auto")}: {n} stage(s) recompute; before the step {base / 2**30:.2f} GiB')
for label, b in eng.memlog:
    print(f'  {label:32s} {b / 2**30:8.2f} GiB')
print(f'  peak allocated {torch.cuda.max_memory_allocated(dev) / 2**30:.2f} GiB; plan: ' +
      ', '.join(f'{k} {v / 2**30:.2f}' for k, v in plan.items() if not isinstance(v, dict) and k != 'recomputing_stages'))
