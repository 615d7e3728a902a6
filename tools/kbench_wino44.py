#!/usr/bin/env python3
"""Launch times of the F(4x4, 3x3) ConvLSTM cell (rnh_wino44_cell), its input transform (rnh_wino44_transform) and the F(2x2, 3x3) cell it would
replace (rnh_conv_wino), at one geometry: HIP events over 200 launches behind 50 warm-ups, each kernel alone on the chip.
    python tools/kbench_wino44.py [N H W]          (default 8 128 128 = BASELINE config 2)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd'))
import torch

from hipvsr.hip_ops import HipOps
from hipvsr.plans import NetPlans, Src
from hipvsr.spec import state_dict_spec
from oracle import refinenet_oracle as orc


def timed(fn, reps=200, warm=50):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8, 128, 128)
    dev = torch.device('cuda:0')
    cfg = orc.exp1_x4_config()
    P, ops = NetPlans(cfg), HipOps(dev)
    spec = state_dict_spec(cfg)
    plan = P.lstm[('forward', 1)]['full']
    g = torch.Generator('cpu').manual_seed(1)
    w, b = (torch.randn(*spec[plan.wkey], generator=g) * 0.03).to(dev), (torch.randn(*spec[plan.bkey], generator=g) * 0.1).to(dev)
    ops.pack(plan, w, b)
    x, h, c = (torch.randn(B, H, W, 64, generator=g).to(dev) for _ in range(3))
    ho, co, go = torch.empty_like(x), torch.empty_like(x), torch.empty(B, H, W, 256, device=dev)
    lstm = dict(hd=64, c_prev=c, h_out=ho, c_out=co, gates_out=go)
    vx, vh = ops.wino44_v(B, H, W, 64)[0], ops.wino44_v(B, H, W, 64)[0]
    ops.wino44_transform(Src(x), B, H, W, vx)
    ops.wino44_transform(Src(h), B, H, W, vh)
    flop = 2.0 * B * H * W * 9 * 128 * 256
    res = {}
    res['F(2x2) cell (rnh_conv_wino)'] = timed(lambda: ops.conv(plan, [Src(x), Src(h)], B, H, W, lstm=lstm))
    res['F(4x4) cell (rnh_wino44_cell)'] = timed(lambda: ops.wino44_cell(plan, [vx, vh], B, H, W, lstm))
    res['input transform, 64 channels (rnh_wino44_transform)'] = timed(lambda: ops.wino44_transform(Src(h), B, H, W, vh))
    res['F(4x4) cell + one transform, back to back'] = timed(lambda: (ops.wino44_cell(plan, [vx, vh], B, H, W, lstm), ops.wino44_transform(Src(ho), B, H, W, vh)))
    print(f'N={B} {H}x{W}, 128 -> 256 columns: direct-form {flop / 1e9:.2f} GFLOP')
    for k, us in res.items():
        print(f'  {k:62s} {us:8.1f} us' + (f'   ({flop / us * 1e-6:7.1f} TFLOP/s direct-form equivalent)' if 'cell' in k else f'   ({(1 + 2.25) * B * H * W * 64 * 4 / us * 1e-6:5.2f} TB/s read + written)'))


if __name__ == '__main__':
    main()
