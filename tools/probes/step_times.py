"""Probe: per-step HIP-event times of bench.py's training step (which steps are the slow ones?).  usage: step_times.py <f32|bf16> [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench, torch
from hipvsr.step_tail import FlatAdam
from src.runner.trainers import AcdcVSRRefineNetTrainer
dt, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device('cuda:0')
args = bench.parse_args([])
net = bench.make_net(dev).set_compute_dtype(dt)
tr = object.__new__(AcdcVSRRefineNetTrainer)
tr.net, tr.loss_fns, tr.metric_fns, tr.optimizer = net, [torch.nn.L1Loss()], [], FlatAdam(net.parameters(), lr=1e-4)
tr.loss_weights = torch.tensor([1.0], device=dev)
tr.graph, tr._graphed = False, None
inputs, targets, pos = bench.synthetic_batch(dev, 8, 7, 128, 128, seed=1)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
import time
host = []
ev[0].record()
for i in range(steps):
    t0 = time.perf_counter()
    tr.train_step(inputs, targets, pos)
    host.append((time.perf_counter() - t0) * 1e3)
    ev[i + 1].record()
torch.cuda.synchronize()
print(dt, 'GPU ms per step:', ' '.join(f'{ev[i].elapsed_time(ev[i + 1]):.1f}' for i in range(steps)))
print(dt, 'host ms to enqueue:', ' '.join(f'{h:.1f}' for h in host))
