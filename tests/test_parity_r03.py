"""Round-3 parity hardening (VERDICT r02 "next" items 2 and 8), on the GPU box, through the C ABI.

* PSNR where the criterion bites: the contract's headline tolerance is |PSNR_build - PSNR_ref| < 0.01 dB (north_star;
  reference src/model/metrics.py:20-36 on outputs denormalised by src/utils.py:14-20, trainer :103-120).  With N(0,1)
  targets a random-init net lands at 15 dB, where 0.01 dB tolerates ~10x the output error it tolerates at the 30+ dB of a
  trained model.  Here the inputs are SURVEY section 8(d)'s structured cine (blurred noise + a beating disc, LR = avg_pool of
  HR) and the targets are the oracle's own fused-group outputs plus noise scaled to ~32 dB - so the criterion is applied at
  the PSNR it is meant for, in fp32 and in the bf16-storage path, at BASELINE config 1 and at config 2's geometry.
* BASELINE configs 4 and 5 at FULL spatial size: x2, T = 5, 256x256 -> 512x512 and x4, T = 11, 96x96 -> 384x384 with
  non-trivial phase codes; N = 1 against the CPU oracle (outputs, loss, every gradient), then the per-GPU batch as
  replicated copies of that sample, which must reproduce it sample by sample, bit for bit (quirk Q8).
* The fused ConvLSTM kernel's gates at config 2's size in BOTH Winograd geometries against float64 (the check that found the
  `nt`-store corruption in round 2, until now a debug script).
"""
import functools
import os

import pytest
import torch

from oracle import refinenet_oracle as orc
from oracle import step_tail_oracle as sto

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def _net(cfg, sd, dtype):
    from src.model.nets import RefineNet
    net = RefineNet(**dict(cfg))
    net.load_state_dict(sd)
    return net.to(_dev()).set_compute_dtype(dtype)


def _trainer(net, loss_fn=None):
    from src.model.metrics import PSNR
    from src.runner.trainers import AcdcVSRRefineNetTrainer
    from src.utils import denormalize
    tr = object.__new__(AcdcVSRRefineNetTrainer)
    tr.net, tr.loss_fns, tr.metric_fns = net, [loss_fn or torch.nn.L1Loss()], [PSNR().to(_dev())]
    tr._denormalize = functools.partial(denormalize, dataset='acdc')
    return tr


# ---------------------------------------------------------------------------------------------------------------------
# PSNR at ~32 dB on the structured cine
# ---------------------------------------------------------------------------------------------------------------------
_PSNR_CASES = {'config 1': (1, 3, 64, 64), 'config 2 geometry': (2, 7, 128, 128)}


@pytest.fixture(scope='module')
def structured_refs():
    """Oracle forward (== reference) on the structured cine, once per geometry; shared by the fp32 and the bf16 test."""
    cache = {}

    def get(name):
        if name not in cache:
            n, t, h, w = _PSNR_CASES[name]
            cfg = orc.exp1_x4_config()
            sd = orc.init_state_dict(cfg, seed=311)
            inputs, hr, pos = orc.structured_cine(cfg, n, t, h, w, seed=312)
            torch.set_num_threads(min(32, os.cpu_count() or 1))
            with torch.no_grad():
                outs = orc.forward(orc.as_leaf_params(sd), cfg, [x.clone() for x in inputs], pos)
            last = [o.detach() for o in outs[-1]]
            # targets: the oracle's fused-group output + white noise of 0.133 normalised units = 6.4 grey levels rms, i.e.
            # PSNR = 10 log10(255^2 / 6.4^2) = 32 dB for an exact implementation
            g = torch.Generator('cpu').manual_seed(313)
            targets = [o + 0.133 * torch.randn(o.shape, generator=g) for o in last]
            cache[name] = dict(cfg=cfg, sd=sd, inputs=inputs, pos=pos, hr=hr, targets=targets, last=last,
                               psnr=float(sto.trainer_metrics(last, targets)[0]))
        return cache[name]
    return get


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('name', list(_PSNR_CASES))
def test_psnr_parity_at_32db_on_structured_cine(structured_refs, name, dtype):
    r = structured_refs(name)
    assert 31.0 < r['psnr'] < 33.0, r['psnr']                    # the criterion is applied where it bites
    net = _net(r['cfg'], r['sd'], dtype)
    tr = _trainer(net)
    dev = _dev()
    net.train()                                                  # the training forward (gates saved): the kernels the bench runs
    outs = net([x.to(dev) for x in r['inputs']], r['pos'].to(dev))
    torch.cuda.synchronize()
    psnr = float(tr._compute_metrics(outs, [t.to(dev) for t in r['targets']])[0])
    worst = max(float((a.detach().cpu() - b).norm()) / float(b.norm()) for a, b in zip(outs[-1], r['last']))
    # per frame as well (the reference's predictor logs PSNR per frame, acdc_vsr_refinenet_predictor.py:67-80)
    per_frame = []
    for a, b, t in zip(outs[-1], r['last'], r['targets']):
        mine = float(sto.trainer_metrics([a.detach().cpu()], [t])[0])
        ref = float(sto.trainer_metrics([b], [t])[0])
        per_frame.append(abs(mine - ref))
    print(f'{name} {dtype}: PSNR {psnr:.4f} vs {r["psnr"]:.4f} dB (oracle), worst per-frame |delta| {max(per_frame):.5f} dB, '
          f'worst relative L2 error of a fused-group output {worst:.2e}')
    assert abs(psnr - r['psnr']) < 0.01, (psnr, r['psnr'])
    assert max(per_frame) < 0.01, per_frame
    # and against the TRUE high-resolution frames (a random-init net is far from them: ~10 dB; still the same criterion)
    hr_psnr = float(tr._compute_metrics(outs, [t.to(dev) for t in r['hr']])[0])
    hr_want = float(sto.trainer_metrics(r['last'], r['hr'])[0])
    assert abs(hr_psnr - hr_want) < 0.01, (hr_psnr, hr_want)


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs 4 and 5 at full spatial size
# ---------------------------------------------------------------------------------------------------------------------
def _grad_close(mine, ref, name, atol=1e-5, rtol=1e-3, l2=1e-3):
    a, b = mine.detach().cpu().double(), ref.detach().cpu().double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    d = (a - b).abs()
    over = d - (atol + rtol * b.abs())
    i = int(over.argmax())
    assert float(over.flatten()[i]) <= 0, (name, 'element', i, float(a.flatten()[i]), float(b.flatten()[i]), 'max|g|', float(b.abs().max()))
    assert float(d.norm()) <= l2 * float(b.norm()) + 1e-12, (name, 'L2', float(d.norm()), float(b.norm()))


def _step(net, inputs, targets, pos):
    dev = _dev()
    tr = _trainer(net)
    net.train()
    outs = net([x.to(dev) for x in inputs], pos.to(dev))
    loss = tr._compute_losses(outs, [t.to(dev) for t in targets])[0]
    net.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    return outs, loss


# name, config overrides, T, H = W, replication for forward + backward, replication for the forward-only batch, bf16 forward + backward batch
# config 4 names batch 16 on one GPU: until round 4 its fp32 training step at N = 16 kept ~270 GB of ConvLSTM states and gates alive and
# did not fit; since the engine's activation-memory plan (hipvsr/engine.py FrameStore, gate recomputation) it does, and
# tests/test_parity_r04.py::test_config4_full_batch_training_step_fp32_and_bf16 runs that step in both precisions.  Here: forward +
# backward at N = 4 and forward-only at N = 8.
# config 5: 32 samples over 4 GPUs = 8 per GPU, forward + backward at the full per-GPU batch.
_FULL = [('cfg4 x2 T=5 256x256', dict(upscale_factor=2), 5, 256, 4, 8, 4),
         ('cfg5 x4 phase code T=11 96x96', dict(), 11, 96, 8, 8, 8)]


@pytest.mark.parametrize('name,over,t,size,n_bwd,n_fwd,n_bf16', _FULL)
def test_full_size_baseline_configs(name, over, t, size, n_bwd, n_fwd, n_bf16, monkeypatch):
    # the per-sample bit-identity below compares runs at N = 1, 4 and 8: one form of the fp32 cell for all of them (F(4x4, 3x3), the default where
    # the images are whole 4x4 tiles - 256 and 96 are -, whatever RNH_WINO44_MIN says)
    monkeypatch.setenv('RNH_WINO44', 'force')
    if size == 256:
        # config 4's full batch keeps the transformed h' in a ring (a slot per frame would take 62 GB) and refine conv1 then runs in F(2x2) form;
        # at N = 1 it would not: that form for every N here (it is what bench.py --config 4 runs)
        monkeypatch.setenv('RNH_WINO44_REFINE', '0')
    cfg = orc.exp1_x4_config(**over)
    sd = orc.init_state_dict(cfg, seed=51)
    inputs, targets, pos = orc.synthetic_batch(cfg, 1, t, size, size, seed=52)
    assert float(pos.std()) > 0.3                                # non-trivial phase codes (config 5's point)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref_out, ref_loss, ref_grads = orc.step(sd, cfg, [x.clone() for x in inputs], targets, pos)
    net = _net(cfg, sd, 'f32')
    # N = 1 against the oracle
    outs1, loss1 = _step(net, inputs, targets, pos)
    worst = 0.0
    for go, gr in zip(outs1, ref_out):
        for a, b in zip(go, gr):
            torch.testing.assert_close(a.detach().cpu(), b, atol=1e-4, rtol=1e-4)
            worst = max(worst, float((a.detach().cpu() - b).abs().max()))
    assert abs(float(loss1.detach()) - float(ref_loss)) <= 1e-5 * abs(float(ref_loss)), (float(loss1), float(ref_loss))
    for k, p in net.named_parameters():
        if ref_grads[k] is None:
            assert p.grad is None
        else:
            _grad_close(p.grad, ref_grads[k], k)
    o1 = [[o.detach().clone() for o in grp] for grp in outs1]
    del outs1
    # forward + backward on n_bwd copies: every sample bit-identical to the oracle-checked one, gradients unchanged (mean over N)
    rep = lambda x, m: torch.cat([x] * m, 0)                               # noqa: E731
    outs, loss = _step(net, [rep(x, n_bwd) for x in inputs], [rep(y, n_bwd) for y in targets], rep(pos, n_bwd))
    for ga, gb in zip(outs, o1):
        for a, b in zip(ga, gb):
            for q in range(n_bwd):
                assert torch.equal(a[q:q + 1], b), (name, 'sample', q)
    assert abs(float(loss) - float(ref_loss)) <= 1e-5 * abs(float(ref_loss))
    for k, p in net.named_parameters():
        if ref_grads[k] is not None:
            _grad_close(p.grad, ref_grads[k], k)
    del outs, loss
    net.zero_grad(set_to_none=True)
    torch.cuda.empty_cache()
    dev = _dev()
    if n_fwd != n_bwd:                                           # a larger batch, forward only
        net.eval()
        with torch.no_grad():
            outs = net([rep(x, n_fwd).to(dev) for x in inputs], rep(pos, n_fwd).to(dev))
        torch.cuda.synchronize()
        for ga, gb in zip(outs, o1):
            for a, b in zip(ga, gb):
                for q in range(n_fwd):
                    assert torch.equal(a[q:q + 1], b), (name, 'forward batch, sample', q)
        del outs
    del net
    torch.cuda.empty_cache()
    if n_bf16:
        # the FULL per-GPU batch through the bf16-storage path, forward AND backward (round 5: until then forward only and switched
        # off, so config 5's bf16 launch geometry - 8 x 23 frames of 96 x 96, 11 supervised - ran nowhere but in bench.py --config 5).
        # N = 1 in bf16 against the oracle under the bf16 criterion; then n_bf16 copies: every sample of all 9 x T outputs bit-identical
        # to the N = 1 run, gradients equal to the N = 1 gradients up to fp32 summation order (the per-sample activation gradients are
        # the N = 1 ones times 1 / n_bf16, a power of two: exact in bf16)
        assert n_bf16 & (n_bf16 - 1) == 0
        nb = _net(cfg, sd, 'bf16')
        outs1, lossb = _step(nb, inputs, targets, pos)
        assert abs(float(lossb) - float(ref_loss)) <= 1e-2 * abs(float(ref_loss)), (float(lossb), float(ref_loss))
        for a, b in zip(outs1[-1], ref_out[-1]):
            assert float((a.detach().cpu() - b).norm()) <= 2e-2 * float(b.norm())
        gb1 = {}
        for k, p in nb.named_parameters():
            if ref_grads[k] is None:
                assert p.grad is None
                continue
            rel = float((p.grad.cpu() - ref_grads[k]).norm()) / float(ref_grads[k].norm())
            assert rel <= 5e-2, (name, 'bf16 N=1', k, rel)
            gb1[k] = p.grad.detach().cpu().clone()
        b1 = [[o.detach().clone() for o in grp] for grp in outs1]
        del outs1
        outs, loss = _step(nb, [rep(x, n_bf16) for x in inputs], [rep(y, n_bf16) for y in targets], rep(pos, n_bf16))
        for ga, gb in zip(outs, b1):
            for a, b in zip(ga, gb):
                for q in range(n_bf16):
                    assert torch.equal(a[q:q + 1], b), (name, 'bf16 full batch, sample', q)
        assert abs(float(loss) - float(lossb)) <= 1e-6 * abs(float(lossb))
        for k, p in nb.named_parameters():
            if k in gb1:
                _grad_close(p.grad, gb1[k], f'{name} bf16 N={n_bf16} {k}')
        del outs, loss, nb
        torch.cuda.empty_cache()
    print(f'{name}: N=1 max |output - oracle| {worst:.2e}, loss {float(loss1):.7f} vs {float(ref_loss):.7f}; N={n_bwd} fwd+bwd and '
          f'N={n_fwd} fwd (bf16 N={n_bf16} fwd+bwd) bit-identical per sample; peak HBM {torch.cuda.max_memory_allocated() / 2**30:.1f} GB')


# ---------------------------------------------------------------------------------------------------------------------
# the fused ConvLSTM cell at config 2's size, both geometries of the Winograd kernel, every value against float64
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('cols', ['64', '128'])
def test_lstm_cell_gates_config2_size_vs_float64(cols):
    """(The round-3 mismatch hunt as a test.)  One ConvLSTM cell launch at N = 8, 128 x 128 - 2048 / 4096 workgroups, several
    of which write disjoint pieces of the same 128-byte lines of the gate tensor - in the 64-column (two workgroups per CU) and
    the 128-column (8-wave) geometry of conv_winoh_kernel: gates, c' and h' of EVERY pixel against a float64 evaluation of
    reference refine_net.py:247-267; no NaN left from the poison fill.  Repeated 5 times (the corruption was intermittent)."""
    import torch.nn.functional as F
    from hipvsr.hip_ops import HipOps
    from hipvsr.plans import NetPlans, Src
    from hipvsr.spec import state_dict_spec
    dev = _dev()
    old = os.environ.get('RNH_WINO_COLS')
    os.environ['RNH_WINO_COLS'] = cols
    try:
        cfg = orc.exp1_x4_config()
        P, ops = NetPlans(cfg), HipOps(dev)
        spec = state_dict_spec(cfg)
        B, H, W = 8, 128, 128
        g = torch.Generator('cpu').manual_seed(7)
        R = lambda *sh: torch.randn(*sh, generator=g)                      # noqa: E731
        pl = P.lstm[('backward', 2)]
        assert getattr(pl['full'], 'wino', False)
        w, b = R(*spec[pl['full'].wkey]) * 0.03, R(*spec[pl['full'].bkey]) * 0.1
        ops.pack(pl['full'], w.to(dev), b.to(dev))
        x, h, c = R(B, H, W, 64), R(B, H, W, 64), R(B, H, W, 64)
        n64 = lambda t: t.double().permute(0, 3, 1, 2)                     # noqa: E731
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        pre = F.conv2d(torch.cat([n64(x), n64(h)], 1), w.double(), b.double(), padding=1)
        gi, gf, gop, gg = pre.split(64, dim=1)
        ref_g = torch.cat([torch.sigmoid(gi), torch.sigmoid(gf), torch.sigmoid(gop), torch.tanh(gg)], 1).permute(0, 2, 3, 1)
        cn = torch.sigmoid(gf) * n64(c) + torch.sigmoid(gi) * torch.tanh(gg)
        hn = (torch.sigmoid(gop) * torch.tanh(cn)).permute(0, 2, 3, 1)
        cn = cn.permute(0, 2, 3, 1)
        xd, hd_, cd = x.to(dev), h.to(dev), c.to(dev)
        for rep in range(5):
            ho, co = (torch.full((B, H, W, 64), float('nan'), device=dev) for _ in range(2))
            go = torch.full((B, H, W, 256), float('nan'), device=dev)
            ops.conv(pl['full'], [Src(xd), Src(hd_)], B, H, W, lstm=dict(hd=64, c_prev=cd, h_out=ho, c_out=co, gates_out=go))
            torch.cuda.synchronize()
            for nm, mine, ref in (('gates', go, ref_g), ('c', co, cn), ('h', ho, hn)):
                m = mine.cpu().double()
                assert not torch.isnan(m).any(), (cols, rep, nm, 'NaN left', int(torch.isnan(m).sum()))
                err = float((m - ref).abs().max())
                assert err <= 1e-4, (cols, rep, nm, err)
    finally:
        if old is None:
            os.environ.pop('RNH_WINO_COLS', None)
        else:
            os.environ['RNH_WINO_COLS'] = old


# ---------------------------------------------------------------------------------------------------------------------
# the two gate-backward kernels: rnh_lstm_gates_bwd_m took over every hd % 8 == 0 call from rnh_lstm_gates_bwd
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('with_state', [True, False])
def test_gates_bwd_m_equals_the_four_element_kernel_bitwise(with_state):
    """On all-fp32 operands the 8-elements-per-thread kernel (csrc/mixed_kernels.hip) and the 4-element kernel it replaced
    (csrc/small_kernels.hip, still serving hd % 8 != 0) evaluate the same expressions in the same order (backward of reference
    refine_net.py:258-265): dgates and dc_prev must agree bit for bit, with and without dc_next / dh2 / c_prev."""
    from hipvsr import lib as L
    from hipvsr.hip_ops import HipOps, _ptr
    dev = _dev()
    ops = HipOps(dev)
    g = torch.Generator('cpu').manual_seed(3)
    npix, hd = 3 * 37 * 29, 64
    R = lambda *sh: torch.randn(*sh, generator=g).to(dev)                  # noqa: E731
    dh, gates, c_next = R(npix, hd), torch.rand(npix, 4 * hd, generator=g).to(dev) * 2 - 0.5, R(npix, hd)
    dh2, dcn, cp = (R(npix, hd), R(npix, hd), R(npix, hd)) if with_state else (None, None, None)
    res = []
    for fn in ('m', 'old'):
        dg, dcp = torch.full((npix, 4 * hd), float('nan'), device=dev), torch.full((npix, hd), float('nan'), device=dev)
        if fn == 'm':
            L.check(ops.lib.rnh_lstm_gates_bwd_m(_ptr(dh), L.DT_F32, _ptr(dh2), L.DT_F32, _ptr(dcn), _ptr(gates), L.DT_F32, _ptr(cp),
                                                 _ptr(c_next), _ptr(dg), L.DT_F32, _ptr(dcp), npix, hd, ops._stream()), 'rnh_lstm_gates_bwd_m')
        else:
            L.check(ops.lib.rnh_lstm_gates_bwd(_ptr(dh), _ptr(dh2), _ptr(dcn), _ptr(gates), _ptr(cp), _ptr(c_next), _ptr(dg), _ptr(dcp),
                                               npix, hd, ops._stream()), 'rnh_lstm_gates_bwd')
        torch.cuda.synchronize()
        assert not torch.isnan(dg).any() and not torch.isnan(dcp).any()
        res.append((dg.cpu(), dcp.cpu()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


# ---------------------------------------------------------------------------------------------------------------------
# refine conv1 side paths of round 3 (fp32): phase-plane weight gradient from border-class sums, 45-tap data-gradient stencil
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('N,T,H,W', [(1, 1, 1, 1), (2, 2, 3, 17), (1, 3, 16, 16), (2, 1, 37, 9), (1, 2, 40, 33)])
def test_refine_side_path_kernels_vs_float64(N, T, H, W):
    """rnh_phase_wgrad (dW of conv1 w.r.t. its five phase-plane input channels, from nine border-class sums per image and channel)
    and rnh_xcol_dgrad (data gradient of conv1's output channel 128: 45 taps per hidden channel) against float64 autograd of the
    convolutions they stand for (reference refine_net.py:149, :176-183), on single-pixel / single-row images, tile and strip tails."""
    import torch.nn.functional as F
    from hipvsr.hip_ops import HipOps
    dev = _dev()
    ops = HipOps(dev)
    g = torch.Generator('cpu').manual_seed(H * 100 + W)
    J, cl, hw = 5, 64, 2
    C1, Cin = 129, 645
    R = lambda *sh: torch.randn(*sh, generator=g)                              # noqa: E731
    w1 = R(C1, Cin, 3, 3) * 0.05
    # --- phase-plane weight gradient: windows i = 0..T-1, planes of frames i + j
    pv = R((T + J - 1) * N)
    P4 = torch.zeros((T + J - 1) * N, H, W, 4)
    P4[..., 0] = pv.view(-1, 1, 1)
    dy = R(T * N, H, W, 132)
    dw = torch.full((C1, Cin, 3, 3), 7.0)
    dwd = dw.clone().to(dev)
    ops.refine_phase_wgrad(dy.to(dev), P4.to(dev), dwd, N, J, cl, 128, False)
    dwd2 = dwd.clone()
    ops.refine_phase_wgrad(dy.to(dev), P4.to(dev), dwd2, N, J, cl, 128, True)
    torch.cuda.synchronize()
    ref = torch.zeros(128, J, 3, 3, dtype=torch.float64)
    for i in range(T):
        x = torch.cat([P4[(i + j) * N:(i + j + 1) * N, ..., :1] for j in range(J)], dim=-1).permute(0, 3, 1, 2).double()
        w0 = torch.zeros(128, J, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv2d(x, w0, padding=1).backward(dy[i * N:(i + 1) * N, ..., :128].permute(0, 3, 1, 2).double())
        ref += w0.grad
    rows = [j * C1 + 2 * cl for j in range(J)]
    mine = dwd.cpu().double()[:128, rows]
    scale = float(ref.abs().max()) + 1e-12
    assert float((mine - ref).abs().max()) <= 2e-5 * scale + 1e-6, float((mine - ref).abs().max())
    assert float((dwd2.cpu().double()[:128, rows] - 2 * ref).abs().max()) <= 4e-5 * scale + 1e-6          # accumulate
    mask = torch.ones(C1, Cin, dtype=torch.bool)
    mask[:128, rows] = False
    assert float((dwd.cpu()[mask] - 7.0).abs().max()) == 0.0                       # everything else untouched
    # --- data gradient of channel 128: gradient planes with hw zero frames on both sides
    gp = torch.zeros((T + 2 * hw) * N, H, W, 132)
    gp[hw * N:(hw + T) * N] = R(T * N, H, W, 132)
    dHf, dHb = R(T * N, H, W, cl), R(T * N, H, W, cl)
    a, b = dHf.clone().to(dev), dHb.clone().to(dev)
    ops.refine_xcol_dgrad(gp.to(dev), w1.to(dev), a, b, N, J, cl)
    torch.cuda.synchronize()
    rf, rb = dHf.double().clone(), dHb.double().clone()
    for f in range(T):
        for j in range(J):
            gj = gp[(f + J - 1 - j) * N:(f + J - j) * N, ..., 128:129].permute(0, 3, 1, 2).double()
            d = F.conv_transpose2d(gj, w1[128:129, j * C1:j * C1 + 2 * cl].double(), padding=1).permute(0, 2, 3, 1)
            rf[f * N:(f + 1) * N] += d[..., :cl]
            rb[f * N:(f + 1) * N] += d[..., cl:]
    for nm, m_, r_ in (('dHf', a, rf), ('dHb', b, rb)):
        e = float((m_.cpu().double() - r_).abs().max())
        assert e <= 2e-5 * float(r_.abs().max()) + 1e-6, (nm, e)
