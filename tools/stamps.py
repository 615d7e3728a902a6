import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd')
sys.path[:0] = [ROOT, PKG]
import shutil
shutil.copy(os.path.join(PKG, 'hipvsr', 'lib_stamps.so'), os.path.join(PKG, 'hipvsr', 'librefinenet_hip.so'))
import torch
from hipvsr.hip_ops import HipOps
from hipvsr.plans import NetPlans, Src
from hipvsr.spec import NetConfig
dev = torch.device('cuda:0')
cfg = NetConfig(1, 1, [64, 64, 64], num_stages=3, refine_window_size=5, upscale_factor=4, update_memory=True, num_updated_frames=6, positional_encoding=True)
P = NetPlans(cfg); ops = HipOps(dev)
N, H, W, T = 8, 128, 128, 7; TN = T * N
pl = P.lstm[('forward', 1)]
xs, hs, gd = torch.randn(TN + N, H, W, 64, device=dev), torch.randn(TN + N, H, W, 64, device=dev), torch.randn(TN, H, W, 256, device=dev)
dw, db = ops.empty(256, 128, 3, 3), ops.empty(256)
for _ in range(3):
    ops.wgrad(pl['wgrad'], [Src(xs, img_off=N), Src(hs)], [Src(gd)], TN, H, W, dw, db)
torch.cuda.synchronize()
z = ops._zero_page.cpu().view(torch.int64)[:4].tolist()
n = z[3]
print('per iteration (cycles): load-issue %.0f  wait %.0f  compute(32 MFMA) %.0f   iterations %d' % (z[0] / n, z[1] / n, z[2] / n, n))
