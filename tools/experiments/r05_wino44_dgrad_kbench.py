"""Isolated launch times of the ConvLSTM cell's data gradient at config 2: F(2x2) (rnh_conv_wino) against transform + F(4x4) (rnh_wino44_transform of the 256-channel gate
gradients, rnh_wino44_conv)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd'))
import torch
from hipvsr.hip_ops import HipOps
from hipvsr.plans import Dst, NetPlans, Src
from hipvsr.spec import state_dict_spec
from oracle import refinenet_oracle as orc


def timed(fn, reps=100, warm=30):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


B, H, W = 8, 128, 128
dev = torch.device('cuda:0')
cfg = orc.exp1_x4_config()
P, ops = NetPlans(cfg), HipOps(dev)
spec = state_dict_spec(cfg)
plan = P.lstm[('forward', 1)]['dgrad']
w = (torch.randn(*spec[plan.wkey]) * 0.03).to(dev)
ops.pack(plan, w)
dg = torch.randn(B, H, W, 256, device=dev)
dx, dh = torch.empty(B, H, W, 64, device=dev), torch.empty(B, H, W, 64, device=dev)
v = ops.wino44_v(B, H, W, 256)[0]
ops.wino44_transform(Src(dg), B, H, W, v)
print('F(2x2) dgrad', round(timed(lambda: ops.conv(plan, [Src(dg)], B, H, W, dsts=[Dst(dx, 64), Dst(dh, 64)])), 1), 'us')
print('transform of 256 channels', round(timed(lambda: ops.wino44_transform(Src(dg), B, H, W, v)), 1), 'us')
print('F(4x4) dgrad on the transformed gradients', round(timed(lambda: ops.wino44_conv(plan, [(v, 0)], B, H, W, [Dst(dx, 64), Dst(dh, 64)])), 1), 'us')
print('transform + F(4x4) dgrad', round(timed(lambda: (ops.wino44_transform(Src(dg), B, H, W, v), ops.wino44_conv(plan, [(v, 0)], B, H, W, [Dst(dx, 64), Dst(dh, 64)]))), 1), 'us')
