#!/usr/bin/env python3
"""Determinism stress: the same training step many times; every output and gradient must be bit-identical from
run to run (no atomics anywhere), so any difference is a race.  GPU box only.  python tests/stress/race_check.py [reps] [N T H]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd')
for p in (ROOT, PKG):
    sys.path.insert(0, p)
import torch                                            # noqa: E402
from oracle import refinenet_oracle as orc              # noqa: E402
from src.model.nets import RefineNet                    # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n, t, h = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (1, 3, 64)
dev = torch.device('cuda:0')
cfg = orc.exp1_x4_config()
sd = orc.init_state_dict(cfg, seed=1)
inputs, targets, pos = orc.synthetic_batch(cfg, n, t, h, h, seed=2)
net = RefineNet(**cfg)
net.load_state_dict(sd)
net = net.to(dev).train()
xs, ys, pc = [x.to(dev) for x in inputs], [y.to(dev) for y in targets], pos.to(dev)
ref = None
bad = {}
for r in range(reps):
    net.zero_grad()
    outs = net(xs, pc)
    loss = sum((o - y).abs().mean() for grp in outs for o, y in zip(grp, ys))
    loss.backward()
    torch.cuda.synchronize()
    cur = {'out%d_%d' % (g, i): o.detach().clone() for g, grp in enumerate(outs) for i, o in enumerate(grp)}
    cur.update({k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
    if ref is None:
        ref = cur
        continue
    for k, v in cur.items():
        if not torch.equal(v, ref[k]):
            d = float((v - ref[k]).abs().max())
            bad.setdefault(k, []).append((r, d))
if os.environ.get('ORACLE'):
    torch.set_num_threads(64)
    with torch.no_grad():
        oo = orc.forward({k: v for k, v in sd.items()}, cfg, [x.clone() for x in inputs], pos)
    for g in range(len(oo)):
        e0 = max(float((ref['out%d_%d' % (g, i)].cpu() - oo[g][i]).abs().max()) for i in range(len(oo[g])))
        e1 = max(float((cur['out%d_%d' % (g, i)].cpu() - oo[g][i]).abs().max()) for i in range(len(oo[g])))
        print('group', g, 'max |hip - oracle|: first run %.3e   last run %.3e' % (e0, e1))
print('reps', reps, 'tensors differing from run 0:', len(bad))
for k, v in list(bad.items())[:30]:
    print(' ', k, 'runs', [a for a, _ in v][:8], 'max diff %.3e' % max(b for _, b in v))
