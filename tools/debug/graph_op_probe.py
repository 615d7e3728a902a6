"""Capture ONE kernel wrapper of the backward in a HIP graph (valid operands), to find which launch the graph path cannot
take.  python tools/debug/graph_op_probe.py <op>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd')
sys.path[:0] = [ROOT, PKG]
import torch
from hipvsr.hip_ops import HipOps
from hipvsr.plans import Dst, NetPlans, Src
from hipvsr.spec import NetConfig, state_dict_spec
op = sys.argv[1]
dev = torch.device('cuda:0')
nf = [64, 64] if 'full' in op else [16, 16]
cfg = NetConfig(1, 1, nf, num_stages=2, refine_window_size=5, upscale_factor=4, update_memory=True, num_updated_frames=2, positional_encoding=True)
P = NetPlans(cfg)
ops = HipOps(dev)
C = nf[0]
params = {k: torch.randn(*s, device=dev) * 0.05 for k, s in state_dict_spec(cfg).items()}
for pl in P.conv_plans():
    ops.pack(pl, params[pl.wkey], params[pl.bkey] if pl.bkey else None)
B, H, W = 2, 16, 32
R = lambda *s: torch.randn(*s, device=dev)
pl = P.lstm[('forward', 1)]
if op.startswith('wgrad'):
    x, h, dy = R(B + 1, H, W, C), R(B + 1, H, W, C), R(B, H, W, 4 * C)
    dw, db = ops.empty(4 * C, 2 * C, 3, 3), ops.empty(4 * C)
    run = lambda: ops.wgrad(pl['wgrad'], [Src(x, img_off=1), Src(h)], [Src(dy)], B, H, W, dw, db)
elif op.startswith('dgrad'):
    dg = R(B, H, W, 4 * C)
    dx, dh = ops.empty(B, H, W, C), ops.empty(B, H, W, C)
    run = lambda: ops.conv(pl['dgrad'], [Src(dg)], B, H, W, dsts=[Dst(dx, C), Dst(dh, C)])
elif op.startswith('gates'):
    t = [R(B, H, W, C) for _ in range(5)]
    g, dg, dcp = torch.sigmoid(R(B, H, W, 4 * C)), ops.empty(B, H, W, 4 * C), ops.empty(B, H, W, C)
    run = lambda: ops.lstm_gates_bwd(t[0], t[1], g, t[2], t[3], dg, dcp, dh2=t[4])
elif op.startswith('uptail'):
    u = P.up[-1]
    y1, d_o = R(B, H, W, C), R(B, 2 * H, 2 * W, 1)
    w2, b2, w3 = params[u['wgrad'].wkey], params[u['wgrad'].bkey], params[P.last_w]
    dw2, db2, dw3, db3 = torch.zeros_like(w2), torch.zeros_like(b2), torch.zeros_like(w3), torch.zeros(1, device=dev)
    def run():
        G = ops.uptail_compose(w2, w3, 2)
        if ops.uptail_xcorr_supported(C, 2, 1):
            M, S = ops.uptail_xcorr(y1, d_o, 2)
        else:
            D = ops.uptail_expand(d_o, 2)
            M, S = ops.empty(P.tail_m.Cout, C, 3, 3), ops.empty(P.tail_m.Cout)
            ops.wgrad(P.tail_m, [Src(y1)], [Src(D)], B, H, W, M, S)
        ops.uptail_wcontract(M, S, w2, b2, w3, dw2, db2, dw3, db3, 2, False, False)
        return ops.uptail_dgrad(d_o, G, C, 2)
elif op.startswith('inconv'):
    x, dy = R(B, H, W, 1), R(B, H, W, C)
    w, b, a = params['in_block.conv.weight'], params['in_block.conv.bias'], params['in_block.prelu.weight']
    dw, db, da = torch.zeros_like(w), torch.zeros_like(b), torch.zeros_like(a)
    run = lambda: ops.inconv_bwd(x, w, b, a, dy, dw, db, da)
elif op.startswith('loss'):
    o, y = R(6, 4096), R(2, 4096)
    gs = torch.ones(6, device=dev)
    run = lambda: ops.loss(o, y, 3, 2, 0, 0.0, gs, want_grad=True)
elif op.startswith('refine'):
    Hf, Hb, P4 = R(B + 4, H, W, C), R(B + 4, H, W, C), R(B + 4, H, W, 4)
    xs = []
    for j in range(5):
        xs += [Src(Hf, img_off=j), Src(Hb, img_off=j), Src(P4, img_off=j)]
    dy = R(B, H, W, P.C1p)
    dw1, db1 = ops.empty(2 * C + 1, 5 * (2 * C + 1), 3, 3), ops.empty(2 * C + 1)
    if P.r1_wino:
        def run():
            ops.wgrad(P.r1_wgrad_h, [s for s in xs if s.t is not P4], [Src(dy, nch=P.r1_cols)], B, H, W, dw1, db1)
            ops.wgrad(P.r1_wgrad_p, [s for s in xs if s.t is P4], [Src(dy, nch=P.r1_cols)], B, H, W, dw1, None, accumulate=True)
            ops.refine_xcol_wgrad([Hf, Hb, P4], dy, dw1, db1, 1, 5, C, True) if False else None
    else:
        run = lambda: ops.wgrad(P.r1_wgrad, xs, [Src(dy, nch=P.r1_cols)], B, H, W, dw1, db1)
if 'side' in op:          # run the op on a side stream forked from / joined to the current one, like the engine's wavefront
    inner, side2 = run, torch.cuda.Stream(dev)
    side3 = torch.cuda.Stream(dev)
    def run():
        ops.fork(4)
        with ops.side(1):
            inner()
            ev = ops.record()
        if 'wait' in op:
            with ops.side(0):
                ops.wait(ev)
                inner()
        ops.join(4)
st = torch.cuda.Stream(dev)
st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    run()
    run()
st.synchronize()
print('warm-up done', op, flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=st):
    run()
print('captured', op, flush=True)
g.replay()
torch.cuda.synchronize()
print('replayed', op, flush=True)
