import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd')
sys.path[:0] = [ROOT, PKG]
import torch
import torch.nn.functional as Fn
from hipvsr.hip_ops import HipOps
from hipvsr.plans import NetPlans, Src, Dst
from hipvsr.spec import NetConfig, state_dict_spec
dev = torch.device('cuda:0')
cfg = NetConfig(1, 1, [64, 64, 64], num_stages=3, refine_window_size=5, upscale_factor=4, update_memory=True, num_updated_frames=6, positional_encoding=True)
P = NetPlans(cfg); ops = HipOps(dev)
torch.manual_seed(0)
params = {k: torch.randn(*s, device=dev) * 0.05 for k, s in state_dict_spec(cfg).items()}
ops.pack(P.r1_fwd_p, params[P.r1_fwd_p.wkey], None)
w1 = params[P.r1_fwd_p.wkey]
widx_p = [j * 129 + 128 for j in range(5)]
for (N, H, W, F) in ((1, 64, 64, 19), (2, 128, 128, 9), (2, 128, 128, 19), (1, 128, 128, 19), (2, 64, 64, 19), (4, 64, 64, 19)):
    P4 = torch.zeros(F * N, H, W, 4, device=dev); P4[..., 0] = torch.randn(F * N, 1, 1, device=dev)
    B = (F - 4) * N
    Rp = torch.zeros((B, H, W, 132), device=dev)
    ops.conv(P.r1_fwd_p, [Src(P4, img_off=j * N) for j in range(5)], B, H, W, dsts=[Dst(Rp, 128, accumulate=True)])
    torch.cuda.synchronize()
    xp = torch.cat([P4[j * N:j * N + B, ..., :1] for j in range(5)], -1)
    ref = Fn.conv2d(xp.permute(0, 3, 1, 2), w1[:128, widx_p], None, padding=1).permute(0, 2, 3, 1)
    d = (Rp[..., :128] - ref).abs()
    bad = (d > 1e-4).any(dim=-1)
    nz = bad.nonzero()
    ys, xs = nz[:, 1], nz[:, 2]
    print((N, H, W, F), 'images', nz[:, 0].unique().tolist()[:12], 'max err %.3e' % float(d.max()), 'bad pixels', int(bad.sum()), 'rows', ys.unique().tolist()[:10], 'cols', xs.unique().tolist()[:10], '..', xs.unique().tolist()[-3:])
