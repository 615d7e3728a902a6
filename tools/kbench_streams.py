#!/usr/bin/env python3
"""What the reversed ConvLSTM wavefront of the bf16 step looks like to the chip: S streams, each a chain of K dependent launches of the
cell's data-gradient kernel (128 columns, N = 8, 128 x 128) - plain STORE epilogue followed by the separate gate backward, or the fused
RNH_EPI_LSTM_BWD epilogue - with independent buffers per stream.  Prints microseconds per (cell, frame).
  python tools/kbench_streams.py [streams=3] [chain=20]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd')
for p in (ROOT, PKG):
    sys.path.insert(0, p)
import torch                                            # noqa: E402
from hipvsr.hip_ops import HipOps                       # noqa: E402
from hipvsr.plans import Dst, NetPlans, Src             # noqa: E402
from hipvsr.spec import NetConfig, state_dict_spec      # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 3
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda:0')
cfg = NetConfig(1, 1, [64, 64, 64], num_stages=3, refine_window_size=5, upscale_factor=4, update_memory=True,
                num_updated_frames=6, positional_encoding=True)
P = NetPlans(cfg, bf16=True)
ops = HipOps(dev)
pl = P.lstm[('forward', 1)]
w = torch.randn(*state_dict_spec(cfg)[pl['full'].wkey], device=dev) * 0.05
ops.pack(pl['dgrad'], w, None)
N, H, W = 8, 128, 128
bf = torch.bfloat16
R = lambda *sh, dtype=bf: (torch.randn(*sh, device=dev) * 0.3).to(dtype)         # noqa: E731
bufs = []
for s in range(S):
    bufs.append(dict(dg=[R(N, H, W, 256), R(N, H, W, 256)], dx=ops.empty(N, H, W, 64, dtype=bf), rec=ops.empty(N, H, W, 64, dtype=bf),
                     dh=R(N, H, W, 64), gates=torch.sigmoid(R(N, H, W, 256).float()).to(bf), cp=R(N, H, W, 64, dtype=torch.float32),
                     cn=R(N, H, W, 64, dtype=torch.float32), dc=[R(N, H, W, 64, dtype=torch.float32), R(N, H, W, 64, dtype=torch.float32)]))
streams = [torch.cuda.Stream(dev) for _ in range(S)]


def run(fused):
    for s, st in enumerate(streams):
        b = bufs[s]
        with torch.cuda.stream(st):
            for k in range(K):
                src, dst = b['dg'][k & 1], b['dg'][(k + 1) & 1]
                if fused:
                    ops.conv(pl['dgrad'], [Src(src)], N, H, W, dsts=[Dst(b['dx'], 64)],
                             lstm_bwd=dict(dh=b['dh'], dc_next=b['dc'][k & 1], gates=b['gates'], c_prev=b['cp'], c_next=b['cn'], dgates=dst,
                                           dc_prev=b['dc'][(k + 1) & 1], hd=64, rec_dtype=bf))
                else:
                    ops.conv(pl['dgrad'], [Src(src)], N, H, W, dsts=[Dst(b['dx'], 64), Dst(b['rec'], 64)])
                    ops.lstm_gates_bwd(b['dh'], b['dc'][k & 1], b['gates'], b['cp'], b['cn'], dst, b['dc'][(k + 1) & 1], dh2=b['rec'])


for fused in (False, True):
    run(fused)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for st in streams:
        st.wait_event(e0)
    for _ in range(3):
        run(fused)
    cur = torch.cuda.current_stream(dev)
    for st in streams:
        ev = torch.cuda.Event()
        ev.record(st)
        cur.wait_event(ev)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (3 * S * K)
    print(f"{'fused' if fused else 'two launches'}: {S} streams x {K} chained cells: {us:7.1f} us per (cell, frame)", flush=True)
