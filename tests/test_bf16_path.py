"""The bf16-storage path (BASELINE.json configs[2]; SURVEY.md section 8c: "bf16: only the PSNR criterion + loss rtol 1e-2").

The reference is fp32 throughout (reference src/model/nets/refine_net.py:234-241 are plain nn.Conv2d), so parity is layered:

* kernel level (GPU): rnh_conv_bf16 / rnh_wgrad_bf16 / the mixed-type helpers against float64 torch evaluations of the SAME
  bf16-rounded operands - what is left is fp32 accumulation order (and, for bf16 destinations, one final rounding);
* engine level (GPU): the HIP engine in bf16 mode against the same engine over the torch double (tests/torch_ops.py), which
  rounds to bf16 where the kernels do, on a ragged full-width shape;
* module level (GPU): the bf16 net against the fp32 CPU oracle (== the reference) under the contract's bf16 criterion:
  |delta PSNR| < 0.01 dB and loss rtol 1e-2, at BASELINE config 1 and at config 2's geometry;
* CPU: tests/test_engine_cpu.py runs the bf16 plans over the torch double against the reference goldens.
"""
import os

import pytest
import torch

from oracle import refinenet_oracle as orc

pytestmark = pytest.mark.gpu

BF16_ULP = 2.0 ** -8          # half a unit in the last place of bf16 (8 significant bits) relative to the value


def _dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def _rb(t):
    """Round to bf16 and back (what a bf16 store or an MFMA operand conversion does)."""
    return t.float().bfloat16().float()


def _close_f32(mine, ref64, name, rel=2e-5):
    """fp32 result of a bf16-operand contraction against float64 on the same operands: accumulation-order noise only."""
    a, b = mine.detach().cpu().double(), ref64.detach().cpu().double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    assert not torch.isnan(a).any(), name
    err = float((a - b).abs().max())
    assert err <= rel * float(b.abs().max()) + 1e-7, (name, err, float(b.abs().max()))


def _close_bf16(mine, ref64, name):
    """bf16 result: the fp32 value rounded once - at most one bf16 ulp of the reference away (rounding boundary flips)."""
    a, b = mine.detach().cpu().double(), ref64.detach().cpu().double()
    assert mine.dtype == torch.bfloat16 and a.shape == b.shape, (name, mine.dtype, a.shape, b.shape)
    over = (a - b).abs() - (2 * BF16_ULP * b.abs() + 2e-5 * float(b.abs().max()) + 1e-30)
    assert float(over.max()) <= 0, (name, float(over.max()), float(b.abs().max()))


def _close(mine, ref64, name):
    (_close_bf16 if mine.dtype == torch.bfloat16 else _close_f32)(mine, ref64, name)


def _full_cfg(**over):
    from hipvsr.spec import NetConfig
    kw = dict(in_channels=1, out_channels=1, num_features=[64, 64, 64], num_stages=3, refine_window_size=5, upscale_factor=4,
              update_memory=True, num_updated_frames=6, positional_encoding=True)
    kw.update(over)
    return NetConfig(**kw)


def _nchw(t):
    return t.detach().cpu().double().permute(0, 3, 1, 2)


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


SHAPES = [(2, 8, 32), (1, 11, 45), (2, 5, 13), (1, 16, 64), (3, 3, 7)]          # (B, H, W): whole tiles, ragged rows and columns


@pytest.mark.parametrize('B,H,W', SHAPES)
@pytest.mark.parametrize('which', ['lstm', 'lstm_first', 'lstm_dgrad', 'up', 'up_f16', 'up_dgrad', 'refine1', 'refine1_dgrad', 'refine2', 'refine2_dgrad',
                                   'refine_1x1'])
def test_conv_bf16_kernel_vs_torch_float64(which, B, H, W):
    import torch.nn.functional as F
    from hipvsr.hip_ops import HipOps
    from hipvsr.plans import Dst, NetPlans, Src
    from hipvsr.spec import state_dict_spec
    dev = _dev()
    cfg = _full_cfg(positional_encoding=which != 'refine_1x1')
    P, ops = NetPlans(cfg, bf16=True), HipOps(dev)
    spec = state_dict_spec(cfg)
    g = torch.Generator('cpu').manual_seed(31 * W + H)
    R = lambda *sh: torch.randn(*sh, generator=g)                           # noqa: E731
    bf = torch.bfloat16
    if which in ('lstm', 'lstm_first', 'lstm_dgrad'):
        pl = P.lstm[('forward', 1)]
        w, b = R(*spec[pl['full'].wkey]) * 0.03, R(256) * 0.1
        if which == 'lstm_dgrad':
            ops.pack(pl['dgrad'], w.to(dev), None)
            dg = R(B, H, W, 256).to(bf)
            dx = torch.full((B, H, W, 64), float('nan'), device=dev, dtype=bf)
            dh = torch.full((B, H, W, 64), float('nan'), device=dev)           # one bf16 and one fp32 destination
            ops.conv(pl['dgrad'], [Src(dg.to(dev))], B, H, W, dsts=[Dst(dx, 64), Dst(dh, 64)])
            torch.cuda.synchronize()
            ref = _nhwc(F.conv_transpose2d(_nchw(dg), _rb(w).double(), padding=1))
            _close(dx, ref[..., :64], 'dx')
            _close(dh, ref[..., 64:], 'dh')
            return
        first = which == 'lstm_first'
        plan = pl['first'] if first else pl['full']
        ops.pack(plan, w.to(dev), b.to(dev))
        x, h, c = R(B + 1, H, W, 64), R(B + 2, H, W, 64).to(bf), R(B, H, W, 64)   # x fp32 (converted on load), h bf16
        ho = torch.full((B, H, W, 64), float('nan'), device=dev, dtype=bf)
        co = torch.full((B, H, W, 64), float('nan'), device=dev)
        go = torch.full((B, H, W, 256), float('nan'), device=dev, dtype=bf)
        srcs = [Src(x.to(dev), img_off=1)] + ([] if first else [Src(h.to(dev), img_off=2)])
        ops.conv(plan, srcs, B, H, W, lstm=dict(hd=64, c_prev=None if first else c.to(dev), h_out=ho, c_out=co, gates_out=go))
        torch.cuda.synchronize()
        xin = _nchw(_rb(x[1:])) if first else torch.cat([_nchw(_rb(x[1:])), _nchw(h[2:])], 1)
        pre = F.conv2d(xin, _rb(w[:, :64] if first else w).double(), b.double(), padding=1)
        gi, gf, gop, gg = pre.split(64, dim=1)
        gi, gf, gop, gg = torch.sigmoid(gi), torch.sigmoid(gf), torch.sigmoid(gop), torch.tanh(gg)
        cn = gi * gg if first else gf * _nchw(c) + gi * gg
        _close(go, _nhwc(torch.cat([gi, gf, gop, gg], 1)), 'gates')
        _close_f32(co, _nhwc(cn), 'c', rel=3e-5)
        _close(ho, _nhwc(gop * torch.tanh(cn)), 'h')
    elif which in ('up', 'up_f16', 'up_dgrad'):
        u = P.up[0]
        w, b = R(256, 64, 3, 3) * 0.04, R(256) * 0.1
        if which == 'up_f16':
            # the f16 MFMA form (round 6: what the engine runs for the upsampler's forward, hipvsr.engine.RefineNetEngine.__init__): IEEE-half
            # weights (rnh_pack_weights_f16), the bf16 input converted exactly - incl. values below half's normal range (2^-14) and a large one;
            # bf16 destination as in the x4 net
            u['fwd'].f16w = True
            ops.pack(u['fwd'], w.to(dev), b.to(dev))
            x = R(B, H, W, 64)
            x[0, 0, 0, :8] = torch.tensor([3e-5, -3e-5, 1e-6, 6.1e-5, 24.0, -40.0, 0.0, 1e-8])
            x = x.to(bf)
            Y = torch.full((B, 2 * H, 2 * W, 64), float('nan'), device=dev)            # (fp32 destination: the contraction itself is checked)
            ops.conv(u['fwd'], [Src(x.to(dev))], B, H, W, ps=(Y, 2))
            torch.cuda.synchronize()
            _close(Y, _nhwc(F.pixel_shuffle(F.conv2d(_nchw(x), w.half().double(), b.double(), padding=1), 2)), 'Y')
            # ... and the form refuses what it cannot take: an fp32 source
            with pytest.raises(Exception, match='f16 weights'):
                ops.conv(u['fwd'], [Src(x.float().to(dev))], B, H, W, ps=(Y, 2))
        elif which == 'up':
            ops.pack(u['fwd'], w.to(dev), b.to(dev))
            x = R(B, H, W, 64).to(bf)
            Y = torch.full((B, 2 * H, 2 * W, 64), float('nan'), device=dev)          # the path keeps this tensor in fp32
            ops.conv(u['fwd'], [Src(x.to(dev))], B, H, W, ps=(Y, 2))
            torch.cuda.synchronize()
            _close(Y, _nhwc(F.pixel_shuffle(F.conv2d(_nchw(x), _rb(w).double(), b.double(), padding=1), 2)), 'Y')
        else:
            ops.pack(u['dgrad'], w.to(dev), None)
            dY = R(B, 2 * H, 2 * W, 64)                                               # fp32 source, pixel-unshuffle gather
            dYd = dY.to(dev)
            dx = torch.full((B, H, W, 64), float('nan'), device=dev, dtype=bf)
            ops.conv(u['dgrad'], [Src(dYd, scale=2, sub=(ij // 2, ij % 2)) for ij in range(4)], B, H, W, dsts=[Dst(dx, 64)])
            torch.cuda.synchronize()
            _close(dx, _nhwc(F.conv_transpose2d(F.pixel_unshuffle(_nchw(_rb(dY)), 2), _rb(w).double(), padding=1)), 'dx')
    elif which in ('refine1', 'refine1_dgrad'):
        w1, b1 = R(129, 645, 3, 3) * 0.02, R(129) * 0.1
        if which == 'refine1':
            ops.pack(P.r1_fwd, w1.to(dev), b1.to(dev))
            Hf, Hb = R(B + 4, H, W, 64).to(bf), R(B + 4, H, W, 64).to(bf)
            P8 = torch.zeros(B + 4, H, W, 8).to(bf)
            P8[..., 0] = R(B + 4, 1, 1).to(bf)
            Hfd, Hbd, P8d = Hf.to(dev), Hb.to(dev), P8.to(dev)
            srcs = []
            for j in range(5):
                srcs += [Src(Hfd, img_off=j), Src(Hbd, img_off=j), Src(P8d, img_off=j)]
            R1 = torch.full((B, H, W, P.C1p), float('nan'), device=dev, dtype=bf)
            ops.conv(P.r1_fwd, srcs, B, H, W, dsts=[Dst(R1, P.r1_cols)])
            torch.cuda.synchronize()
            xin = torch.cat([torch.cat([_nchw(Hf[j:j + B]), _nchw(Hb[j:j + B]), _nchw(P8[j:j + B, ..., :1])], 1) for j in range(5)], 1)
            ref = _nhwc(F.conv2d(xin, _rb(w1).double(), b1.double(), padding=1))
            _close(R1[..., :129], ref, 'R1')
            assert float(R1[..., 129:].float().abs().max()) == 0.0                # the pad columns are written as zeros
        else:
            ops.pack(P.r1_dgrad, w1.to(dev), None)
            gs = torch.zeros(B + 4, H, W, P.C1p)
            gs[..., :129] = R(B + 4, H, W, 129)
            gs = gs.to(bf)
            base_f, base_b = R(B, H, W, 64).to(bf), R(B, H, W, 64)
            dHf, dHb = base_f.to(dev), base_b.to(dev)
            ops.conv(P.r1_dgrad, [Src(gs.to(dev), img_off=4 - j) for j in range(5)], B, H, W,
                     dsts=[Dst(dHf, 64, accumulate=True), Dst(dHb, 64, accumulate=True)])
            torch.cuda.synchronize()
            tot = 0
            for j in range(5):
                tot = tot + F.conv_transpose2d(_nchw(gs[4 - j:4 - j + B, ..., :129]), _rb(w1[:, j * 129:j * 129 + 128]).double(), padding=1)
            tot = _nhwc(tot)
            _close(dHf, base_f.double() + tot[..., :64], 'dHf')
            _close(dHb, base_b.double() + tot[..., 64:], 'dHb')
    elif which in ('refine2', 'refine2_dgrad'):
        w2, b2 = R(64, 129, 3, 3) * 0.05, R(64) * 0.1
        if which == 'refine2':
            ops.pack(P.r2_fwd, w2.to(dev), b2.to(dev))
            R1 = torch.zeros(B, H, W, P.C1p)
            R1[..., :129] = R(B, H, W, 129)
            R1[..., 129:] = 7.0                                                     # pad channels must not contribute (zero weights)
            R1 = R1.to(bf)
            out = torch.full((B, H, W, 64), float('nan'), device=dev, dtype=bf)
            ops.conv(P.r2_fwd, [Src(R1.to(dev))], B, H, W, dsts=[Dst(out, 64)])
            torch.cuda.synchronize()
            _close(out, _nhwc(F.conv2d(_nchw(R1[..., :129]), _rb(w2).double(), b2.double(), padding=1)), 'R')
        else:
            ops.pack(P.r2_dgrad, w2.to(dev), None)
            dR = R(B, H, W, 64).to(bf)
            out = torch.full((B + 2, H, W, P.C1p), float('nan'), device=dev, dtype=bf)
            ops.conv(P.r2_dgrad, [Src(dR.to(dev))], B, H, W, dsts=[Dst(out, P.C1p, img_off=1)])
            torch.cuda.synchronize()
            ref = _nhwc(F.conv_transpose2d(_nchw(dR), _rb(w2).double(), padding=1))
            _close(out[1:1 + B, ..., :129], ref, 'dR1')
            assert float(out[1:1 + B, ..., 129:].float().abs().max()) == 0.0
            assert bool(torch.isnan(out[0].float()).all()) and bool(torch.isnan(out[-1].float()).all())      # neighbours untouched
    else:
        w1, b1 = R(64, 640, 1, 1) * 0.05, R(64) * 0.1
        ops.pack(P.r1_fwd, w1.to(dev), b1.to(dev))
        Hf, Hb = R(B + 4, H, W, 64).to(bf), R(B + 4, H, W, 64).to(bf)
        Hfd, Hbd = Hf.to(dev), Hb.to(dev)
        srcs = []
        for j in range(5):
            srcs += [Src(Hfd, img_off=j), Src(Hbd, img_off=j)]
        out = torch.full((B, H, W, 64), float('nan'), device=dev, dtype=bf)
        ops.conv(P.r1_fwd, srcs, B, H, W, dsts=[Dst(out, 64)])
        torch.cuda.synchronize()
        xin = torch.cat([torch.cat([_nchw(Hf[j:j + B]), _nchw(Hb[j:j + B])], 1) for j in range(5)], 1)
        _close(out, _nhwc(F.conv2d(xin, _rb(w1).double(), b1.double())), 'R(1x1)')


@pytest.mark.parametrize('B,H,W', [(3, 6, 32), (2, 40, 45), (2, 5, 13), (1, 70, 64)])
@pytest.mark.parametrize('which', ['lstm', 'up', 'refine1', 'refine2'])
def test_wgrad_bf16_kernel_vs_torch_float64(which, B, H, W):
    """rnh_wgrad_bf16 + rnh_wgrad_reduce against float64 autograd of conv2d on the bf16-rounded operands: row strips with
    several work items per workgroup (H > 32), ragged strips (W % 32 != 0), fp32 and bf16 sources, the pixel-unshuffle
    gather of the dy operand, padded rows / columns (8-channel phase planes with one real channel, 136 -> 129), bias."""
    import torch.nn.functional as F
    from hipvsr.hip_ops import HipOps
    from hipvsr.plans import NetPlans, Src
    dev = _dev()
    P, ops = NetPlans(_full_cfg(), bf16=True), HipOps(dev)
    g = torch.Generator('cpu').manual_seed(7 + W + H)
    R = lambda *sh: torch.randn(*sh, generator=g)                           # noqa: E731
    bf = torch.bfloat16

    def ref_wgrad(x_nchw, dy_nchw, cout, cin):
        w0 = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv2d(x_nchw, w0, padding=1).backward(dy_nchw)
        return w0.grad, dy_nchw.sum(dim=(0, 2, 3))

    if which == 'lstm':
        plan = P.lstm[('backward', 0)]['wgrad']
        x, h, dy = R(B + 1, H, W, 64).to(bf), R(B + 1, H, W, 64).to(bf), R(B, H, W, 256).to(bf)
        xs, ys = [Src(x.to(dev), img_off=1), Src(h.to(dev))], [Src(dy.to(dev))]
        shape = (256, 128, 3, 3)
        rw, rb = ref_wgrad(torch.cat([_nchw(x[1:]), _nchw(h[:B])], 1), _nchw(dy), 256, 128)
    elif which == 'up':
        plan = P.up[0]['wgrad']
        x, big = R(B, H, W, 64).to(bf), R(B, 2 * H, 2 * W, 64)                # dy: fp32, gathered from the 2x larger tensor
        xs = [Src(x.to(dev))]
        bigd = big.to(dev)
        ys = [Src(bigd, scale=2, sub=(ij // 2, ij % 2)) for ij in range(4)]
        shape = (256, 64, 3, 3)
        rw, rb = ref_wgrad(_nchw(x), F.pixel_unshuffle(_nchw(_rb(big)), 2), 256, 64)
    elif which == 'refine1':
        plan = P.r1_wgrad
        Hf, Hb = R(B + 4, H, W, 64).to(bf), R(B + 4, H, W, 64).to(bf)
        P8 = torch.zeros(B + 4, H, W, 8)
        P8[..., 0] = R(B + 4, 1, 1)
        P8[..., 1:] = 3.0                                                       # pad channels of the plane: rows the plan does not map
        P8 = P8.to(bf)
        dy = torch.zeros(B, H, W, P.C1p)
        dy[..., :129] = R(B, H, W, 129)
        dy[..., 129:] = 5.0                                                     # pad columns: not mapped either
        dy = dy.to(bf)
        Hfd, Hbd, P8d = Hf.to(dev), Hb.to(dev), P8.to(dev)
        xs = []
        for j in range(5):
            xs += [Src(Hfd, img_off=j), Src(Hbd, img_off=j), Src(P8d, img_off=j)]
        ys = [Src(dy.to(dev), nch=P.r1_cols)]
        shape = (129, 645, 3, 3)
        xin = torch.cat([torch.cat([_nchw(Hf[j:j + B]), _nchw(Hb[j:j + B]), _nchw(P8[j:j + B, ..., :1])], 1) for j in range(5)], 1)
        rw, rb = ref_wgrad(xin, _nchw(dy[..., :129]), 129, 645)
    else:
        plan = P.r2_wgrad
        R1 = torch.zeros(B + 1, H, W, P.C1p)
        R1[..., :129] = R(B + 1, H, W, 129)
        R1 = R1.to(bf)
        dy = R(B, H, W, 64).to(bf)
        xs, ys = [Src(R1.to(dev), img_off=1)], [Src(dy.to(dev))]
        shape = (64, 129, 3, 3)
        rw, rb = ref_wgrad(_nchw(R1[1:, ..., :129]), _nchw(dy), 64, 129)
    dw, db = torch.full(shape, float('nan'), device=dev), torch.full(shape[:1], float('nan'), device=dev)
    ops.wgrad(plan, xs, ys, B, H, W, dw, db)
    torch.cuda.synchronize()
    _close_f32(dw, rw, f'{which}.dw', rel=3e-5)
    _close_f32(db, rb, f'{which}.db', rel=3e-5)
    dw2, db2 = dw.clone(), db.clone()
    ops.wgrad(plan, xs, ys, B, H, W, dw2, db2, accumulate=True)               # accumulate: exactly twice; and bitwise repeatable
    torch.cuda.synchronize()
    assert torch.equal(dw2, dw + dw) and torch.equal(db2, db + db)
    # the register-staged kernel (RNH_WGRAD_DMA=0, read per call) against the LDS-DMA kernel that serves bf16 x bf16 sources by default:
    # the same products, partitioned over the workgroups differently (row strips vs contiguous step ranges) - fp32 re-association only
    old = os.environ.get('RNH_WGRAD_DMA')
    os.environ['RNH_WGRAD_DMA'] = '0'
    try:
        dw3, db3 = torch.full(shape, float('nan'), device=dev), torch.full(shape[:1], float('nan'), device=dev)
        ops.wgrad(plan, xs, ys, B, H, W, dw3, db3)
        torch.cuda.synchronize()
    finally:
        if old is None:
            os.environ.pop('RNH_WGRAD_DMA', None)
        else:
            os.environ['RNH_WGRAD_DMA'] = old
    _close_f32(dw3, rw, f'{which}.dw[register-staged]', rel=3e-5)
    _close_f32(db3, rb, f'{which}.db[register-staged]', rel=3e-5)
    assert float((dw3 - dw).abs().max()) <= 3e-5 * float(rw.abs().max()) and float((db3 - db).abs().max()) <= 3e-5 * float(rb.abs().max())


def test_mixed_type_helpers_vs_torch():
    from hipvsr.hip_ops import HipOps
    from torch_ops import TorchOps
    dev = _dev()
    ops, ref = HipOps(dev), TorchOps('cpu')
    g = torch.Generator('cpu').manual_seed(3)
    R = lambda *sh: torch.randn(*sh, generator=g)                           # noqa: E731
    bf = torch.bfloat16
    # add: every combination of operand types, store and accumulate
    a, b, c = R(3, 5, 7, 16), R(3, 5, 7, 16).to(bf), R(3, 5, 7, 16)
    for odt in (torch.float32, bf):
        for acc in (False, True):
            o0 = R(3, 5, 7, 16).to(odt)
            od = o0.to(dev)
            ops.add(od, a.to(dev), b.to(dev), c.to(dev), accumulate=acc)
            want = ref.add(o0.clone(), a, b, c, accumulate=acc)
            torch.cuda.synchronize()
            assert torch.equal(od.cpu(), want), (odt, acc)
    # cast both ways; phase plane
    x = R(2, 4, 4, 24)
    assert torch.equal(ops.cast(x.to(dev), bf).cpu(), x.to(bf)) and torch.equal(ops.cast(x.to(bf).to(dev), torch.float32).cpu(), x.to(bf).float())
    pos = torch.rand(3, 5, 1, generator=g) * 2 - 1
    pp = ops.phase_plane(pos.to(dev), 3, 5, 4, 6, dtype=bf, channels=8)
    assert torch.equal(pp.cpu(), ref.phase_plane(pos, 3, 5, 4, 6, dtype=bf, channels=8))
    # gate backward: bf16 dh / gates / dgates, fp32 cell states; with and without the optional operands
    hd, npix = 16, 3 * 5 * 7
    for has in (True, False):
        dh, dh2 = R(npix, hd).to(bf), (R(npix, hd) if has else None)           # dh2 may have another type than dh
        gates = torch.sigmoid(R(npix, 4 * hd)).to(bf)
        cn, cp, dcn = R(npix, hd), (R(npix, hd) if has else None), (R(npix, hd) if has else None)
        dgd, dcpd = torch.full((npix, 4 * hd), float('nan'), device=dev, dtype=bf), torch.full((npix, hd), float('nan'), device=dev)
        D = lambda t: None if t is None else t.to(dev)                      # noqa: E731
        ops.lstm_gates_bwd(D(dh), D(dcn), D(gates), D(cp), D(cn), dgd, dcpd, dh2=D(dh2))
        dg_ref, dcp_ref = torch.zeros(npix, 4 * hd, dtype=bf), torch.zeros(npix, hd)
        ref.lstm_gates_bwd(dh, dcn, gates, cp, cn, dg_ref, dcp_ref, dh2=dh2)
        torch.cuda.synchronize()
        torch.testing.assert_close(dcpd.cpu(), dcp_ref, atol=1e-6, rtol=1e-5)
        torch.testing.assert_close(dgd.cpu().float(), dg_ref.float(), atol=1e-6, rtol=2 ** -7)   # one rounding boundary at most


def test_engine_bf16_vs_torch_double_on_a_ragged_shape():
    """The whole bf16 engine (every launch of the path: 64-column and 128-column tiles, all three epilogues, multi-source K,
    accumulating stores, weight gradients) against the same engine over the torch double, which rounds to bf16 at the same
    places.  What differs is accumulation order - and the occasional bf16 rounding boundary that a 1e-7 difference flips,
    whose 4e-3 relative step then propagates: hence L2 criteria, not elementwise ones."""
    from hipvsr.engine import RefineNetEngine
    from hipvsr.hip_ops import HipOps
    from hipvsr.spec import NetConfig
    from torch_ops import TorchOps
    dev = _dev()
    kw = dict(in_channels=1, out_channels=1, num_features=[64, 64], num_stages=2, refine_window_size=5, upscale_factor=4,
              update_memory=True, num_updated_frames=2, positional_encoding=True)
    cfg = NetConfig(**kw)
    sd = orc.init_state_dict(orc.Config(**kw), seed=3)
    inputs, targets, pos = orc.synthetic_batch(orc.Config(**kw), n=2, t=2, h=20, w=13, seed=4)
    res = {}
    for name, ops, d in (('hip', HipOps(dev), dev), ('ref', TorchOps('cpu'), torch.device('cpu'))):
        eng = RefineNetEngine(cfg, ops, dtype='bf16')
        params = {k: v.to(d) for k, v in sd.items()}
        O, ctx = eng.forward(params, [x.to(d) for x in inputs], pos.to(d), need_grad=True)
        g = torch.Generator('cpu').manual_seed(9)
        dO = (torch.randn(O.shape, generator=g) * 1e-3).to(d)
        grads = eng.backward(params, ctx, dO)
        res[name] = (O.cpu(), {k: (v.cpu() if v is not None else None) for k, v in grads.items()})
    a, b = res['hip'][0].double(), res['ref'][0].double()
    assert not torch.isnan(a).any()
    assert float((a - b).norm()) <= 3e-3 * float(b.norm()), (float((a - b).norm()), float(b.norm()))
    for k, v in res['ref'][1].items():
        if v is None:
            assert res['hip'][1][k] is None
            continue
        mine = res['hip'][1][k].double()
        assert not torch.isnan(mine).any(), k
        assert float((mine - v.double()).norm()) <= 1e-2 * float(v.double().norm()) + 1e-9, (k, float((mine - v.double()).norm()), float(v.norm()))


def _module_step(kwargs, sd, inputs, targets, pos, dtype):
    from src.model.nets import RefineNet
    from src.runner.trainers import AcdcVSRRefineNetTrainer
    dev = _dev()
    net = RefineNet(**kwargs)
    net.load_state_dict(sd)
    net = net.to(dev).set_compute_dtype(dtype)
    tr = object.__new__(AcdcVSRRefineNetTrainer)
    tr.net, tr.loss_fns, tr.metric_fns = net, [torch.nn.L1Loss()], []
    net.train()
    outs = net([x.to(dev) for x in inputs], pos.to(dev))
    loss = tr._compute_losses(outs, [t.to(dev) for t in targets])[0]
    net.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    return net, tr, outs, loss


def _psnr(tr, outs, targets):
    import functools
    from src.model.metrics import PSNR
    from src.utils import denormalize
    tr.metric_fns = [PSNR().to(_dev())]
    tr._denormalize = functools.partial(denormalize, dataset='acdc')
    return float(tr._compute_metrics(outs, [t.to(_dev()) for t in targets])[0])


@pytest.mark.parametrize('name,n,t,h,w', [('BASELINE config 1', 1, 3, 64, 64), ('config 2 / 3 geometry', 2, 7, 128, 128)])
def test_bf16_module_vs_fp32_oracle(name, n, t, h, w):
    """The contract's bf16 criterion against the CPU oracle (== the reference) on identical inputs:
    |delta PSNR| < 0.01 dB and loss rtol 1e-2; gradients (no contract figure for bf16) within 5 % in L2 per tensor.
    Also: the state_dict is untouched by the dtype switch and the outputs / parameter gradients are fp32."""
    from oracle import step_tail_oracle as sto
    cfg = orc.exp1_x4_config()
    sd = orc.init_state_dict(cfg, seed=77)
    inputs, targets, pos = orc.synthetic_batch(cfg, n, t, h, w, seed=78)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref_out, ref_loss, ref_grads = orc.step(sd, cfg, [x.clone() for x in inputs], targets, pos)
    net, tr, outs, loss = _module_step(dict(cfg), sd, inputs, targets, pos, 'bf16')
    assert net._engine().bf16 and all(o.dtype == torch.float32 for grp in outs for o in grp)
    for k, v in net.state_dict().items():
        assert v.dtype == torch.float32 and torch.equal(v.cpu(), sd[k]), k
    assert abs(float(loss.detach()) - float(ref_loss)) <= 1e-2 * abs(float(ref_loss)), (float(loss), float(ref_loss))
    psnr, want = _psnr(tr, outs, targets), float(sto.trainer_metrics(ref_out[-1], targets)[0])
    assert abs(psnr - want) < 0.01, (psnr, want)
    worst = 0.0
    for go, gr in zip(outs, ref_out):
        for a, b in zip(go, gr):
            worst = max(worst, float((a.detach().cpu() - b).norm()) / float(b.norm()))
    assert worst <= 2e-2, worst
    gw = 0.0
    for k, p in net.named_parameters():
        if ref_grads[k] is None:
            assert p.grad is None
            continue
        assert p.grad.dtype == torch.float32
        rel = float((p.grad.cpu() - ref_grads[k]).norm()) / float(ref_grads[k].norm())
        gw = max(gw, rel)
        assert rel <= 5e-2, (k, rel)
    print(f'bf16 vs fp32 oracle, {name}: loss {float(loss):.6f} vs {float(ref_loss):.6f}, PSNR {psnr:.4f} vs {want:.4f}, '
          f'worst relative L2 error of an output {worst:.2e}, of a gradient {gw:.2e}')


def test_bf16_bench_launch_geometry_n8_equals_replicated_n2():
    """The bf16 twin of tests/test_hip_parity.py::test_bench_launch_geometry_n8_equals_replicated_n2 - the exact launch
    geometry of `bench.py`'s `secondary` line and of BASELINE config 3 per GPU (N = 8, T = 7, 128x128, bf16 storage): a
    batch made of four copies of the N = 2 batch that test_bf16_module_vs_fp32_oracle checks against the oracle.  Samples
    are independent bit for bit (quirk Q8; every tile of a bf16 launch belongs to one image), so each of the 8 samples of
    each of the 63 outputs equals its N = 2 twin exactly - which pins the 2048-workgroup cell launches, the 45-window refine
    launches and the 168-image upsampler launches.  Backward: the loss is a mean over samples, so every per-sample activation
    gradient is the N = 2 one times 1/4 - a power of two, exact in bf16 - and the parameter gradients are equal up to the
    summation order of the fp32 accumulators, which depends on N through the weight gradients' split counts
    (hipvsr/plans.py WgradPlan.nsplit_bf16): the fp32 contract's criterion (elementwise 1e-5 + 1e-3 |g|, L2 1e-3) holds between
    the two bf16 runs."""
    cfg = orc.exp1_x4_config()
    sd = orc.init_state_dict(cfg, seed=77)
    inputs, targets, pos = orc.synthetic_batch(cfg, 2, 7, 128, 128, seed=78)     # the batch of test_bf16_module_vs_fp32_oracle[config 2 / 3]
    net2, _, outs2, loss2 = _module_step(dict(cfg), sd, inputs, targets, pos, 'bf16')
    g2 = {k: p.grad.detach().cpu().clone() for k, p in net2.named_parameters() if p.grad is not None}
    o2 = [[o.detach().clone() for o in grp] for grp in outs2]
    del net2, outs2
    rep = lambda t: torch.cat([t] * 4, 0)                                   # noqa: E731
    net8, _, outs8, loss8 = _module_step(dict(cfg), sd, [rep(x) for x in inputs], [rep(t) for t in targets], rep(pos), 'bf16')
    assert net8._engine().bf16
    assert len(outs8) == 9 and all(len(grp) == 7 for grp in outs8)
    for ga, gb in zip(outs8, o2):
        for a, b in zip(ga, gb):
            assert tuple(a.shape) == (8, 1, 512, 512)
            for q in range(4):
                assert torch.equal(a[2 * q:2 * q + 2], b), q
    assert abs(float(loss8) - float(loss2)) <= 1e-6 * abs(float(loss2))
    worst = 0.0
    for k, p in net8.named_parameters():
        if k not in g2:
            assert p.grad is None, k
            continue
        a, b = p.grad.detach().cpu().double(), g2[k].double()
        d = (a - b).abs()
        assert float((d - (1e-5 + 1e-3 * b.abs())).max()) <= 0, (k, float(d.max()), float(b.abs().max()))
        rel = float(d.norm()) / float(b.norm())
        worst = max(worst, rel)
        assert rel <= 1e-3, (k, rel)
    print(f'bf16 N=8 bench geometry: 63 outputs bit-identical per sample to the N=2 run; worst relative L2 difference of a gradient {worst:.2e}')


G1_CASES = [f'x{s}_pos{p}_mem{m}' for s in (2, 3, 4) for p in (1, 0) for m in (1, 0)] + ['x8_pos1_mem1']


@pytest.mark.parametrize('case', G1_CASES)
def test_bf16_module_on_the_reference_goldens(golden_dir, case):
    """Every variant of the net the reference's goldens cover - x2 / x3 / x4 / x8 upsamplers (one, two and three PixelShuffle
    stages: the tail input is Sb or an inner feature map), with / without the phase code (3x3 or 1x1 refine conv1), with /
    without memory, 8-channel layers (one 8-channel group of a 128-column LSTM tile) - through the bf16 kernels, against the
    reference's own fp32 outputs, loss and gradients: bf16 criterion (loss rtol 1e-2; L2 2 % on outputs, 6 % on gradients)."""
    c = torch.load(os.path.join(golden_dir, 'g1_tiny.pt'), weights_only=False)[case]
    net, tr, outs, loss = _module_step(c['kwargs'], c['state_dict'], c['inputs'], c['targets'], c['pos_codes'], 'bf16')
    assert len(outs) == len(c['outputs'])
    for go, gr in zip(outs, c['outputs']):
        for a, b in zip(go, gr):
            assert a.shape == b.shape and a.dtype == torch.float32
            assert float((a.detach().cpu() - b).norm()) <= 2e-2 * float(b.norm()) + 1e-3
    assert abs(float(loss.detach()) - float(c['train_loss'])) <= 1e-2 * abs(float(c['train_loss']))
    for k, p in net.named_parameters():
        if c['grads'][k] is None:
            assert p.grad is None, k
        else:
            d = float((p.grad.cpu() - c['grads'][k]).norm())
            assert d <= 6e-2 * float(c['grads'][k].norm()) + 1e-6, (k, d, float(c['grads'][k].norm()))
    net.eval()
    with torch.no_grad():
        last = net([x.to(_dev()) for x in c['inputs']], c['pos_codes'].to(_dev()))[-1]
    for a, b in zip(last, c['eval_last']):
        assert float((a.cpu() - b).norm()) <= 2e-2 * float(b.norm()) + 1e-3


def test_bf16_training_step_is_bitwise_repeatable_and_switchable():
    """No atomics in the bf16 kernels either (fixed-order slab reduction): the same step gives the same bits; switching the
    module back to 'f32' rebuilds the engine and reproduces the fp32 result bit for bit."""
    from src.model.nets import RefineNet
    dev = _dev()
    cfg = orc.exp1_x4_config(num_updated_frames=3)
    net = RefineNet(**cfg)
    net.load_state_dict(orc.init_state_dict(cfg, seed=1))
    net = net.to(dev).train()
    inputs, targets, pos = orc.synthetic_batch(cfg, 2, 2, 40, 45, seed=2)
    xs, ys, pc = [x.to(dev) for x in inputs], [y.to(dev) for y in targets], pos.to(dev)

    def step():
        net.zero_grad()
        outs = net(xs, pc)
        sum((o - y).abs().mean() for grp in outs for o, y in zip(grp, ys)).backward()
        torch.cuda.synchronize()
        return [o.detach().clone() for grp in outs for o in grp] + [p.grad.clone() for p in net.parameters() if p.grad is not None]

    f32 = step()
    net.set_compute_dtype('bf16')
    ref = step()
    assert not all(torch.equal(a, b) for a, b in zip(ref, f32))
    for r in range(5):
        assert all(torch.equal(a, b) for a, b in zip(step(), ref)), r
    net.set_compute_dtype('f32')
    assert all(torch.equal(a, b) for a, b in zip(step(), f32))
    with pytest.raises(ValueError):
        net.set_compute_dtype('fp8')


# ---------------------------------------------------------------------------------------------------------------------
# the collapsed upsampler tail with bf16 input / input gradient (csrc/uptail_bf16.hip; round 3)
# ---------------------------------------------------------------------------------------------------------------------
_TAIL_SHAPES = [(2, 1, 1), (1, 1, 7), (1, 5, 1), (3, 19, 37), (1, 16, 32), (2, 33, 16), (1, 8, 100), (2, 64, 64)]


def _tail_inputs(B, Hm, Wm, seed):
    g = torch.Generator('cpu').manual_seed(seed)
    Cq = 64
    y1 = torch.randn(B, Hm, Wm, 64, generator=g).bfloat16()                 # what the PixelShuffle epilogue stores
    w2 = torch.randn(Cq * 4, 64, 3, 3, generator=g) * 0.05
    b2 = torch.randn(Cq * 4, generator=g) * 0.1
    w3 = torch.randn(1, Cq, 3, 3, generator=g) * 0.05
    b3 = torch.randn(1, generator=g)
    d_o = torch.randn(B, 2 * Hm, 2 * Wm, 1, generator=g)
    return y1, w2, b2, w3, b3, d_o


@pytest.mark.parametrize('B,Hm,Wm', _TAIL_SHAPES)
def test_bf16_tail_forward_vs_float64(B, Hm, Wm):
    """rnh_uptail_fwd_bf16 (composed 5x5 convolution on v_mfma_f32_16x16x32_f16 - since round 6 the composed weights are IEEE half and the bf16
    input converts exactly -, four output rows per accumulator tile, border paths subtracted) against conv2d -> pixel_shuffle -> conv2d in
    float64 on the SAME bf16 input (reference refine_net.py:199-205): single-row / single-column images, tile tails in both directions, several
    tiles per workgroup.  What is left is the rounding of the composed weights to half (2^-12 relative per weight, 1600 terms): 5e-4 of the
    output's largest magnitude (with bf16 weights, until round 5: 4e-3); and against the fp32 tail kernel on the same input, likewise."""
    import torch.nn.functional as F
    from hipvsr.hip_ops import HipOps
    dev = _dev()
    ops = HipOps(dev)
    y1, w2, b2, w3, b3, _ = _tail_inputs(B, Hm, Wm, 100 + Hm)
    ref = F.conv2d(F.pixel_shuffle(F.conv2d(y1.double().permute(0, 3, 1, 2), w2.double(), b2.double(), padding=1), 2),
                   w3.double(), b3.double(), padding=1).permute(0, 2, 3, 1)
    out = torch.full((B, 2 * Hm, 2 * Wm, 1), float('nan'), device=dev)
    ops.uptail_fwd(y1.to(dev), w2.to(dev), b2.to(dev), w3.to(dev), b3.to(dev), 2, out)
    out32 = torch.full((B, 2 * Hm, 2 * Wm, 1), float('nan'), device=dev)
    ops.uptail_fwd(y1.float().to(dev), w2.to(dev), b2.to(dev), w3.to(dev), b3.to(dev), 2, out32)
    torch.cuda.synchronize()
    assert not torch.isnan(out).any()
    scale = float(ref.abs().max())
    err = float((out.cpu().double() - ref).abs().max())
    l2 = float((out.cpu().double() - ref).norm()) / float(ref.norm())
    assert err <= 5e-4 * scale and l2 <= 2.5e-4, (err, scale, l2)
    assert float((out32.cpu().double() - ref).abs().max()) <= 1e-4 * scale           # (the fp32 kernel on the same input)


@pytest.mark.parametrize('B,Hm,Wm', _TAIL_SHAPES)
def test_bf16_tail_backward_kernels_vs_float64(B, Hm, Wm):
    """rnh_uptail_dgrad_bf16 (dY1 in bf16) and rnh_uptail_xcorr_bf16 (M, S from a bf16 Y1) against the layer-by-layer
    float64 backward (autograd of reference refine_net.py:199-205).  d_o enters as a bf16 hi + lo pair (16 bits), Y1 is exact
    (it IS bf16): M and S are limited by fp32 accumulation only; dY1 by the bf16 rounding of the composed weights and of the
    stored result (2 x 2^-9)."""
    from hipvsr.hip_ops import HipOps
    from torch_ops import TorchOps
    dev = _dev()
    ops, ref = HipOps(dev), TorchOps('cpu')
    y1, w2, _, w3, _, d_o = _tail_inputs(B, Hm, Wm, 200 + Wm)
    G = ops.uptail_compose(w2.to(dev), w3.to(dev), 2)
    dy1 = ops.uptail_dgrad(d_o.to(dev), G, 64, 2, dtype=torch.bfloat16)
    M, S = ops.uptail_xcorr(y1.to(dev), d_o.to(dev), 2)
    torch.cuda.synchronize()
    assert dy1.dtype == torch.bfloat16 and not torch.isnan(dy1.float()).any()
    want = ref.uptail_dgrad(d_o.double(), dict(w2=w2.double(), w3=w3.double(), r=2), 64, 2)
    scale = float(want.abs().max())
    err = float((dy1.cpu().double() - want).abs().max())
    l2 = float((dy1.cpu().double() - want).norm()) / float(want.norm())
    assert err <= 8e-3 * scale and l2 <= 4e-3, ('dY1', err, scale, l2)
    ref64 = TorchOps('cpu')
    Mr, Sr = ref64.uptail_xcorr(y1.double(), d_o.double(), 2)
    for nm, mine, r_ in (('M', M, Mr), ('S', S, Sr)):
        e = float((mine.cpu().double() - r_.double()).abs().max())
        assert e <= 3e-5 * float(r_.abs().max()) + 1e-6, (nm, e, float(r_.abs().max()))


@pytest.mark.parametrize('out_dt', [torch.bfloat16, torch.float32])
def test_xcol_m_kernels_vs_torch(out_dt):
    """rnh_xcol_combine_m / rnh_xcol_gather_m (the frame-wise side path of refine conv1's last channel in the bf16-storage path,
    reference refine_net.py:147-151, :176-183) against the torch double: exact (sums of at most 5 fp32 terms, one rounding)."""
    from hipvsr.hip_ops import HipOps
    from torch_ops import TorchOps
    dev = _dev()
    ops, ref = HipOps(dev), TorchOps('cpu')
    g = torch.Generator('cpu').manual_seed(31)
    N, J, nwin, H, W, C, c0 = 2, 5, 3, 7, 9, 136, 128
    z = torch.randn((nwin + J - 1) * N, H, W, 8, generator=g)
    b1 = torch.randn(C, generator=g)
    R1 = torch.randn(nwin * N, H, W, C, generator=g).to(out_dt)
    R1d = R1.clone().to(dev)
    ops.xcol_combine_m(z.to(dev), b1.to(dev), R1d, N, J, c0)
    ref.xcol_combine_m(z, b1, R1, N, J, c0)
    torch.cuda.synchronize()
    assert torch.equal(R1d.cpu()[..., :c0], R1[..., :c0])                          # other channels untouched
    torch.testing.assert_close(R1d.cpu().float()[..., c0:], R1.float()[..., c0:], atol=1e-6 if out_dt == torch.float32 else 0, rtol=1e-6 if out_dt == torch.float32 else 8e-3)
    dy = torch.randn(nwin * N, H, W, C, generator=g).to(out_dt)
    E = ops.xcol_gather_m(dy.to(dev), N, J, c0, out_dt)
    Er = ref.xcol_gather_m(dy, N, J, c0, out_dt)
    torch.cuda.synchronize()
    assert E.dtype == out_dt and torch.equal(E.cpu(), Er)


@pytest.mark.gpu
@pytest.mark.parametrize('B,H,W', [(2, 8, 32), (1, 11, 45), (2, 5, 13), (1, 24, 64)])
@pytest.mark.parametrize('nf,dh_dt,rec_dt,edge', [(64, torch.bfloat16, torch.bfloat16, 'mid'), (64, torch.float32, torch.bfloat16, 'first'),
                                                  (64, torch.bfloat16, torch.float32, 'last'), (8, torch.bfloat16, torch.bfloat16, 'mid'),
                                                  (32, torch.float32, torch.float32, 'mid')])
def test_fused_gate_backward_epilogue_equals_the_two_launches_bitwise(nf, dh_dt, rec_dt, edge, B, H, W, monkeypatch):
    """rnh_conv_bf16 with RNH_EPI_LSTM_BWD (the data gradient of a ConvLSTM cell with the gate backward of the chain's next frame in its
    epilogue) against what it replaces: the same convolution storing [input gradient | recurrent state gradient] followed by
    rnh_lstm_gates_bwd_m - bit for bit, in the 128-column (64 + 64) and the 64-column tile (8 + 8, 32 + 32), on whole and ragged tiles,
    with and without c_prev / dc_next / dc_prev, fp32 and bf16 summands.  (The two-launch path itself is held against float64 by
    test_conv_bf16_kernel_vs_torch_float64[lstm_dgrad] and test_mixed_type_helpers_vs_torch.)"""
    from hipvsr.hip_ops import HipOps
    from hipvsr.plans import Dst, NetPlans, Src
    from hipvsr.spec import state_dict_spec
    monkeypatch.setenv('RNH_FUSE_GATES_BWD', '1')                              # (the suite may run with the A/B switch set the other way)
    dev = _dev()
    cfg = _full_cfg(num_features=[nf, nf])
    P, ops = NetPlans(cfg, bf16=True), HipOps(dev)
    pl = P.lstm[('backward', 1)]
    hd, cx = pl['hd'], pl['cx']
    assert ops.lstm_bwd_fusable(pl['dgrad'], cx, hd)
    spec = state_dict_spec(cfg)
    g = torch.Generator('cpu').manual_seed(7 * W + H + nf)
    R = lambda *sh: torch.randn(*sh, generator=g)                           # noqa: E731
    bf = torch.bfloat16
    ops.pack(pl['dgrad'], (R(*spec[pl['full'].wkey]) * 0.05).to(dev), None)
    dg_t = R(B, H, W, 4 * hd).to(bf).to(dev)                                  # gate gradients of frame t (the convolution's input)
    dh = (R(B, H, W, hd) * 0.5).to(dh_dt).to(dev)
    gates = torch.sigmoid(R(B, H, W, 4 * hd)).to(bf).to(dev)
    c_prev = None if edge == 'first' else R(B, H, W, hd).to(dev)
    c_next = R(B, H, W, hd).to(dev)
    dc_next = None if edge == 'first' else (R(B, H, W, hd) * 0.5).to(dev)
    nan = lambda *sh, dt=torch.float32: torch.full(sh, float('nan'), device=dev, dtype=dt)      # noqa: E731
    # two launches
    dx_a, rec = nan(B, H, W, cx, dt=bf), nan(B, H, W, hd, dt=rec_dt)
    dgates_a, dcp_a = nan(B, H, W, 4 * hd, dt=bf), (None if edge == 'last' else nan(B, H, W, hd))
    ops.conv(pl['dgrad'], [Src(dg_t)], B, H, W, dsts=[Dst(dx_a, cx), Dst(rec, hd)])
    ops.lstm_gates_bwd(dh, dc_next, gates, c_prev, c_next, dgates_a, dcp_a, dh2=rec)
    # one launch
    dx_b, dgates_b, dcp_b = nan(B, H, W, cx, dt=bf), nan(B, H, W, 4 * hd, dt=bf), (None if edge == 'last' else nan(B, H, W, hd))
    ops.conv(pl['dgrad'], [Src(dg_t)], B, H, W, dsts=[Dst(dx_b, cx)],
             lstm_bwd=dict(dh=dh, dc_next=dc_next, gates=gates, c_prev=c_prev, c_next=c_next, dgates=dgates_b, dc_prev=dcp_b, hd=hd, rec_dtype=rec_dt))
    torch.cuda.synchronize()
    assert not torch.isnan(dgates_b.float()).any() and not torch.isnan(dx_b.float()).any()
    assert torch.equal(dx_a, dx_b)
    assert torch.equal(dgates_a, dgates_b), (dgates_a.float() - dgates_b.float()).abs().max()
    if dcp_a is not None:
        assert torch.equal(dcp_a, dcp_b), (dcp_a - dcp_b).abs().max()


@pytest.mark.gpu
def test_fused_and_two_launch_training_steps_agree_bitwise(monkeypatch):
    """The bf16-storage training step with the gate backward fused into the data-gradient launches (default) and with
    RNH_FUSE_GATES_BWD=0 (a launch of its own per cell and frame): outputs and every gradient bit-identical, at full width, on a
    ragged shape, T = 4 supervised frames (chains of head + 3 fused launches)."""
    cfg = orc.exp1_x4_config(num_updated_frames=2)
    sd = orc.init_state_dict(cfg, seed=5)
    inputs, targets, pos = orc.synthetic_batch(cfg, 2, 4, 24, 40, seed=3)
    res = {}
    for flag in ('1', '0'):
        monkeypatch.setenv('RNH_FUSE_GATES_BWD', flag)
        net, _, outs, loss = _module_step(dict(cfg), sd, inputs, targets, pos, 'bf16')
        res[flag] = (loss.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
    assert torch.equal(res['1'][0], res['0'][0])
    assert res['1'][1].keys() == res['0'][1].keys() and len(res['1'][1]) > 20
    for k in res['1'][1]:
        assert torch.equal(res['1'][1][k], res['0'][1][k]), k
