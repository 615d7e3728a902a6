#!/usr/bin/env python3
"""Training trajectories at the reference YAML's training shape (batch 16, 32x32 crops, T = 7, Adam lr 1e-4, L1) on the structured cine: fp32,
fp32 from an initialisation perturbed by 1e-6 relative (the noise floor of a chaotic trajectory), bf16-storage.  Prints 25-step loss windows and
the validation PSNR against the true HR frames every 50 steps.  usage: train_traj.py [steps]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd'))
import torch                                  # noqa: E402
import test_parity_r04 as T4                  # noqa: E402
from oracle import refinenet_oracle as orc    # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
cfg = orc.exp1_x4_config()
c = T4._TRAIN
sd0 = orc.init_state_dict(cfg, seed=61)
dev = torch.device('cuda:0')
pool = orc.structured_cine(cfg, c['pool'], c['t'], c['crop'], c['crop'], seed=62)
pin, ptg, ppos = [x.to(dev) for x in pool[0]], [y.to(dev) for y in pool[1]], pool[2].to(dev)
val = orc.structured_cine(cfg, 4, c['t'], 64, 64, seed=63)
vin, vtg, vpos = [x.to(dev) for x in val[0]], [y.to(dev) for y in val[1]], val[2].to(dev)
nb = c['pool'] // c['batch']
res = {}
for name, dt, eps in (('f32', 'f32', 0.0), ('f32 perturbed', 'f32', 1e-6), ('bf16', 'bf16', 0.0)):
    g = torch.Generator().manual_seed(5)
    sd = {k: v * (1 + eps * torch.randn(v.shape, generator=g)) for k, v in sd0.items()}
    net = T4._net(cfg, sd, dt).train()
    tr = T4._train_trainer(net, c['lr'])
    losses, psnrs = [], []
    for i in range(steps):
        sl = slice((i % nb) * c['batch'], (i % nb + 1) * c['batch'])
        _, loss, _ = tr.train_step([x[sl] for x in pin], [y[sl] for y in ptg], ppos[sl])
        losses.append(loss.detach())
        if (i + 1) % 50 == 0:
            net.eval()
            with torch.no_grad():
                psnrs.append(float(tr._compute_metrics(net(vin, vpos), vtg)[0]))
            net.train()
    losses = [float(x) for x in losses]
    res[name] = ([sum(losses[i:i + 25]) / 25 for i in range(0, steps, 25)], psnrs)
    print(name, 'loss windows', ' '.join(f'{x:.4f}' for x in res[name][0]))
    print(name, 'PSNR', ' '.join(f'{x:.3f}' for x in psnrs), flush=True)
for a in ('f32 perturbed', 'bf16'):
    w = [abs(x - y) / x for x, y in zip(res['f32'][0], res[a][0])]
    p = [abs(x - y) for x, y in zip(res['f32'][1], res[a][1])]
    print(f'{a} vs f32: loss windows rel diff ' + ' '.join(f'{x:.1e}' for x in w))
    print(f'{a} vs f32: |dPSNR| ' + ' '.join(f'{x:.3f}' for x in p))
