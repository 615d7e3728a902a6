#!/usr/bin/env python3
"""VERDICT r04 item 5, decided on the CPU before any HIP is written: would a Winograd F(4x4,3x3) ConvLSTM cell FORWARD in true fp32 arithmetic keep
the contract's parity criteria?  The oracle's training step (== the reference) is run three times on identical inputs - with the cell's
convolution (a) direct, (b) in F(2x2,3x3) form (what csrc/conv_wino.hip computes), (c) in F(4x4,3x3) form - transforms and the 16 / 36
element-wise GEMMs in fp32 with fp32 accumulation, the backward in every case the exact one (the prototype the verdict asks for replaces the
forward cell launch only).  Reported against the bars of tests/test_hip_parity.py::test_cfg2_geometry_vs_oracle: outputs atol = rtol = 1e-4, loss
rtol 1e-5, gradients elementwise 1e-5 + 1e-3 |g| and 1e-3 in L2, |delta PSNR| < 0.01 dB.  A float64 run of (a) gives the scale of fp32 noise.
    python tools/wino43_study.py [N T H W]          (default 1 3 64 64 = BASELINE config 1; "2 7 128 128" = config 2's geometry, ~10 min)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

from oracle import refinenet_oracle as orc
from oracle import step_tail_oracle as sto

BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)
BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def wino_conv(x, w, b, m):
    """3x3, padding 1, in Winograd F(m x m, 3x3) form; every step in x.dtype (fp32: transforms as matrix products, the GEMMs as one einsum)."""
    BT, G, AT = (BT4, G4, AT4) if m == 4 else (BT2, G2, AT2)
    BT, G, AT = BT.to(x.dtype), G.to(x.dtype), AT.to(x.dtype)
    a = m + 2
    N, C, H, W = x.shape
    th, tw = (H + m - 1) // m, (W + m - 1) // m
    xp = F.pad(x, (1, 1 + tw * m - W, 1, 1 + th * m - H))
    d = xp.unfold(2, a, m).unfold(3, a, m)                          # N C th tw a a
    V = torch.einsum('ij,nctujk,lk->nctuil', BT, d, BT)             # B^T d B
    U = torch.einsum('ij,ocjk,lk->ocil', G, w, G)                   # G g G^T
    M = torch.einsum('ocil,nctuil->notuil', U, V)
    Y = torch.einsum('ij,notujk,lk->notuil', AT, M, AT)             # N O th tw m m
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(N, w.shape[0], th * m, tw * m)[:, :, :H, :W]
    return y + b.view(1, -1, 1, 1)


def run(sd, cfg, inputs, targets, pos, mode, dtype=torch.float32):
    real = F.conv2d

    def conv(x, w, b=None, **kw):
        if mode and w.shape[1] == 2 * cfg.num_features[0] and w.shape[0] == 4 * cfg.num_features[0] and w.shape[-1] == 3:
            exact = real(x, w, b, **kw)
            return exact + (wino_conv(x, w, b, mode) - exact).detach()          # forward value: Winograd; gradient: the exact one
        return real(x, w, b, **kw)

    orc.F.conv2d = conv
    try:
        cast = lambda t: t.to(dtype)                                             # noqa: E731
        p = {k: cast(v).clone().requires_grad_(True) for k, v in sd.items()}     # (orc.step with leaves of `dtype`)
        outs = orc.forward(p, cfg, [cast(x) for x in inputs], cast(pos))
        loss = orc.training_loss(outs, [cast(t) for t in targets])
        loss.backward()
        grads = {k: (v.grad.detach().clone() if v.grad is not None else None) for k, v in p.items()}
        out, loss = tuple([o.detach() for o in grp] for grp in outs), loss.detach()
    finally:
        orc.F.conv2d = real
    return out, loss, grads


def main():
    n, t, h, w = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (1, 3, 64, 64)
    torch.set_num_threads(os.cpu_count() or 1)
    cfg = orc.exp1_x4_config()
    sd = orc.init_state_dict(cfg, seed=202)
    inputs, targets, pos = orc.synthetic_batch(cfg, n, t, h, w, seed=203)
    # self-check of the two transforms in float64
    x, wt, b = torch.randn(1, 8, 13, 10, dtype=torch.float64), torch.randn(5, 8, 3, 3, dtype=torch.float64), torch.randn(5, dtype=torch.float64)
    for m in (2, 4):
        assert float((wino_conv(x, wt, b, m) - F.conv2d(x, wt, b, padding=1)).abs().max()) < 1e-11, m
    ref = run(sd, cfg, inputs, targets, pos, 0)
    r64 = run(sd, cfg, inputs, targets, pos, 0, torch.float64)
    print(f'N={n} T={t} {h}x{w}: oracle loss {float(ref[1]):.7f}')
    psnr_ref = float(sto.trainer_metrics(ref[0][-1], targets)[0])
    for name, res in (('fp32 direct vs its float64 run (fp32 noise floor)', r64), ('F(2x2,3x3) cell forward', run(sd, cfg, inputs, targets, pos, 2)),
                      ('F(4x4,3x3) cell forward', run(sd, cfg, inputs, targets, pos, 4))):
        out, loss, grads = res
        worst_o, over_o = 0.0, 0.0
        for ga, gb in zip(out, ref[0]):
            for a, bb in zip(ga, gb):
                d = (a.double() - bb.double()).abs()
                worst_o = max(worst_o, float(d.max()))
                over_o = max(over_o, float((d / (1e-4 + 1e-4 * bb.double().abs())).max()))
        worst_g, worst_l2, kg = 0.0, 0.0, ''
        for k, g in ref[2].items():
            if g is None:
                continue
            d = (grads[k].double() - g.double()).abs()
            r_el = float((d / (1e-5 + 1e-3 * g.double().abs())).max())
            r_l2 = float(d.norm() / g.double().norm()) / 1e-3
            if max(r_el, r_l2) > max(worst_g, worst_l2):
                kg = k
            worst_g, worst_l2 = max(worst_g, r_el), max(worst_l2, r_l2)
        dl = abs(float(loss) - float(ref[1])) / abs(float(ref[1]))
        dps = abs(float(sto.trainer_metrics([o.float() for o in out[-1]], targets)[0]) - psnr_ref)
        print(f'  {name:52s} outputs: max |d| {worst_o:.2e} = {over_o:.3f} of the bar; loss rel {dl:.1e} = {dl / 1e-5:.3f} of the bar; gradients: '
              f'{worst_g:.3f} (elementwise) / {worst_l2:.3f} (L2) of the bar, worst {kg}; |dPSNR| {dps:.1e} dB')


if __name__ == '__main__':
    main()
