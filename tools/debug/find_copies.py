"""Where do the device-to-device copies of a training step come from?  (torch profiler with stacks)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd')
sys.path[:0] = [ROOT, PKG]
import torch
from bench import make_net, synthetic_batch
from hipvsr.step_tail import FlatAdam
from src.runner.trainers import AcdcVSRRefineNetTrainer
dev = torch.device('cuda:0')
net = make_net(dev)
tr = object.__new__(AcdcVSRRefineNetTrainer)
tr.net, tr.loss_fns, tr.metric_fns, tr.optimizer = net, [torch.nn.L1Loss()], [], FlatAdam(net.parameters(), lr=1e-4)
tr.loss_weights = torch.tensor([1.0], device=dev)
inputs, targets, pos = synthetic_batch(dev, 8, 7, 128, 128, seed=1)
for _ in range(2):
    tr.train_step(inputs, targets, pos)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    tr.train_step(inputs, targets, pos)
    torch.cuda.synchronize()
import collections
agg = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::copy_', 'aten::contiguous', 'aten::clone', 'aten::zeros', 'aten::fill_', 'aten::zero_', 'aten::mul', 'aten::stack', 'aten::cat', 'aten::to', 'aten::_to_copy'):
        st = [f for f in (ev.stack or []) if 'repo' in f or 'hipvsr' in f or 'src/' in f]
        key = (ev.name, str(ev.input_shapes)[:60], st[0][-90:] if st else '?')
        agg[key] += 1
for k, v in agg.most_common(40):
    print(v, k)
