"""rnh_wino44f_wgrad_v vs rnh_wino44f_wgrad vs the F(2x2)-tile kernel, with and without a synchronisation between the calls (diagnosis, round 6)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd'), os.path.join(ROOT, 'tests')]
import torch
from hipvsr.hip_ops import HipOps
from hipvsr.plans import NetPlans, Src
from oracle import refinenet_oracle as orc
dev = torch.device('cuda:0')
P, ops = NetPlans(orc.exp1_x4_config()), HipOps(dev)
plan = P.lstm[('backward', 2)]['wgrad']
for rep, (vN, nfr, H, W, sync) in enumerate([(4, 1, 128, 128, False), (8, 1, 128, 128, False), (8, 1, 128, 128, True), (8, 2, 128, 128, False), (8, 2, 128, 128, True), (8, 1, 128, 128, False)]):
    B = vN * nfr
    g = torch.Generator('cpu').manual_seed(1)
    x, h, dy = (torch.randn(B, H, W, c, generator=g).to(dev) for c in (64, 64, 256))
    Vx, Vh = ops.wino44_v(vN, H, W, 64, frames=nfr), ops.wino44_v(vN, H, W, 64, frames=nfr)
    for f in range(nfr):
        ops.wino44_transform(Src(x, img_off=f * vN), vN, H, W, Vx[f])
        ops.wino44_transform(Src(h, img_off=f * vN), vN, H, W, Vh[f])
    dws = [torch.zeros(256, 128, 3, 3, device=dev) for _ in range(4)]
    dbs = [torch.zeros(256, device=dev) for _ in range(4)]
    os.environ['RNH_WINO44F_WGRAD'] = 'all'
    ops.wgrad(plan, [Src(x), Src(h)], [Src(dy)], B, H, W, dws[0], dbs[0], vsrcs=[(Vx, 0, 1), (Vh, 0, 1)], vN=vN)
    if sync:
        torch.cuda.synchronize()
    ops.wgrad(plan, [Src(x), Src(h)], [Src(dy)], B, H, W, dws[1], dbs[1])
    if sync:
        torch.cuda.synchronize()
    ops.wgrad(plan, [Src(x), Src(h)], [Src(dy)], B, H, W, dws[2], dbs[2], vsrcs=[(Vx, 0, 1), (Vh, 0, 1)], vN=vN)
    os.environ['RNH_WINO44F_WGRAD'] = '0'
    ops.wgrad(plan, [Src(x), Src(h)], [Src(dy)], B, H, W, dws[3], dbs[3])
    torch.cuda.synchronize()
    sc = float(dws[3].abs().max())
    print(f'vN {vN} frames {nfr} sync {sync}: |V - F22| {float((dws[0] - dws[3]).abs().max()) / sc:.2e}  |raw - F22| {float((dws[1] - dws[3]).abs().max()) / sc:.2e}  '
          f'|V again - F22| {float((dws[2] - dws[3]).abs().max()) / sc:.2e}', flush=True)
