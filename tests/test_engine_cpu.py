"""Host logic on CPU: the engine's forward/backward scheduling and the GEMM index maps of hipvsr.plans, run
over the torch test double (tests/torch_ops.py), against the golden vectors captured from the reference.

This does NOT exercise the HIP kernels (tests/test_hip_*.py do, on the GPU box); it proves that what the
engine asks the kernels to do is the reference's computation (reference src/model/nets/refine_net.py:61-135
and the backward autograd derives from it).  Tolerance: fp32 re-association only (1e-5 abs on O(1) values).
"""
import os

import numpy as np
import pytest
import torch

from hipvsr import lib as L
from hipvsr.engine import RefineNetEngine
from hipvsr.spec import NetConfig, state_dict_spec
from torch_ops import TorchOps


@pytest.fixture(scope='module')
def g1(golden_dir):
    return torch.load(os.path.join(golden_dir, 'g1_tiny.pt'), weights_only=False)


def run_engine(c, kind=L.LOSS_L1, eps=1e-6, dtype='f32'):
    cfg = NetConfig(**c['kwargs'])
    ops = TorchOps('cpu')
    eng = RefineNetEngine(cfg, ops, dtype=dtype)
    params = {k: v.clone() for k, v in c['state_dict'].items()}
    O_all, ctx = eng.forward(params, c['inputs'], c['pos_codes'], need_grad=True)
    S, T = cfg.num_stages, len(c['targets'])
    N = c['inputs'][0].shape[0]
    G = 3 * S
    y = torch.stack(c['targets'], 0).permute(0, 1, 3, 4, 2).contiguous()        # (T, N, sH, sW, Co)
    gscale = torch.tensor([np.power(0.5, S - 1 - g // 3) / T for g in range(G) for _ in range(T)], dtype=torch.float32)
    o = O_all.reshape(G * T, *O_all.shape[3:]) if False else O_all.reshape(G, T, N, *O_all.shape[3:])
    losses, dO = ops.loss(o.reshape(G * T, -1), y.reshape(T, -1), G, T, kind, eps, gscale, want_grad=True)
    total = (losses * gscale).sum()
    grads = eng.backward(params, ctx, dO.reshape(O_all.shape))
    return cfg, O_all, total, grads


CASES = [f'x{s}_pos{p}_mem{m}' for s in (2, 3, 4) for p in (1, 0) for m in (1, 0)] + ['x8_pos1_mem1']


@pytest.mark.parametrize('case', CASES)
def test_engine_matches_reference_golden(g1, case):
    c = g1[case]
    cfg, O_all, total, grads = run_engine(c)
    S, T = cfg.num_stages, len(c['targets'])
    N = c['inputs'][0].shape[0]
    assert list(grads.keys()) == list(state_dict_spec(cfg).keys())
    for g in range(3 * S):
        for i in range(T):
            mine = O_all[g // 3, g % 3, i * N:(i + 1) * N].permute(0, 3, 1, 2)
            torch.testing.assert_close(mine, c['outputs'][g][i], atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(total, c['train_loss'], atol=1e-5, rtol=1e-5)
    for k, gref in c['grads'].items():
        if gref is None:
            assert grads[k] is None
            continue
        assert not torch.isnan(grads[k]).any(), k
        scale = float(gref.abs().max()) + 1e-12
        err = float((grads[k] - gref).abs().max())
        assert err <= 2e-4 * scale + 1e-7, (k, err, scale)


@pytest.mark.parametrize('case', CASES)
def test_engine_bf16_storage_plans_match_reference_golden(g1, case):
    """The bf16-storage path's plans (no Winograd, no side paths, 8-channel phase planes, 64-column tiles) and dtype
    plumbing, over the torch double that rounds to bf16 where the HIP kernels do: same computation as the reference up
    to bf16 rounding of the stored activations and MFMA operands (8 mantissa bits: relative 4e-3 per rounding)."""
    c = g1[case]
    cfg, O_all, total, grads = run_engine(c, dtype='bf16')
    S, T = cfg.num_stages, len(c['targets'])
    N = c['inputs'][0].shape[0]
    assert O_all.dtype == torch.float32
    for g in range(3 * S):
        for i in range(T):
            mine = O_all[g // 3, g % 3, i * N:(i + 1) * N].permute(0, 3, 1, 2)
            ref = c['outputs'][g][i]
            assert float((mine - ref).norm()) <= 2e-2 * float(ref.norm()) + 1e-3, (g, i)
    assert abs(float(total) - float(c['train_loss'])) <= 1e-2 * abs(float(c['train_loss']))
    for k, gref in c['grads'].items():
        if gref is None:
            assert grads[k] is None
            continue
        assert grads[k].dtype == torch.float32 and not torch.isnan(grads[k]).any(), k
        assert float((grads[k] - gref).norm()) <= 6e-2 * float(gref.norm()) + 1e-6, (k, float((grads[k] - gref).norm()), float(gref.norm()))


def test_engine_charbonnier(g1):
    c = g1['x4_pos1_mem1']
    _, _, total, grads = run_engine(c, kind=L.LOSS_CHARBONNIER)
    torch.testing.assert_close(total, c['charbonnier_train_loss'], atol=1e-5, rtol=1e-5)
    for k, gref in c['charbonnier_grads'].items():
        if gref is not None:
            scale = float(gref.abs().max()) + 1e-12
            assert float((grads[k] - gref).abs().max()) <= 2e-4 * scale + 1e-7, k


def test_engine_flat_gradient_buffer(g1):
    c = g1['x2_pos1_mem1']
    cfg = NetConfig(**c['kwargs'])
    ops = TorchOps('cpu')
    eng = RefineNetEngine(cfg, ops)
    params = {k: v.clone() for k, v in c['state_dict'].items()}
    O_all, ctx = eng.forward(params, c['inputs'], c['pos_codes'], need_grad=True)
    n = sum(int(np.prod(s)) for s in state_dict_spec(cfg).values())
    flat = torch.zeros(n)
    grads = eng.backward(params, ctx, torch.ones_like(O_all), flat=flat)
    off = 0
    for k, shp in state_dict_spec(cfg).items():
        m = int(np.prod(shp))
        if grads[k] is not None:
            assert grads[k].data_ptr() == flat.data_ptr() + 4 * off
            assert torch.equal(grads[k].reshape(-1), flat[off:off + m])
        off += m


def test_engine_errors():
    base = dict(in_channels=1, out_channels=1, num_features=[8, 8])
    with pytest.raises(ValueError, match='upscale factor'):
        NetConfig(upscale_factor=5, **base)
    with pytest.raises(ValueError, match='update_memory'):
        NetConfig(num_updated_frames=2, update_memory=False, **base)
    cfg = NetConfig(num_stages=2, update_memory=True, num_updated_frames=0, positional_encoding=True, **base)
    eng = RefineNetEngine(cfg, TorchOps('cpu'))
    with pytest.raises(IndexError):
        eng.forward({}, [torch.zeros(1, 1, 4, 4)] * 6, torch.zeros(1, 6, 1), need_grad=False)


def test_engine_odd_channel_side_path_vs_oracle():
    """num_features 32 => refine conv1 has 65 = 2*32 + 1 output channels: the GEMM plans cover 64 columns and the last
    channel goes through refine_xcol_fwd / refine_xcol_wgrad.  Engine over the torch double against the oracle."""
    from oracle import refinenet_oracle as orc
    kw = dict(in_channels=1, out_channels=1, num_features=[32, 32], num_stages=2, refine_window_size=5, upscale_factor=2,
              update_memory=True, num_updated_frames=2, positional_encoding=True)
    ocfg = orc.Config(**kw)
    sd = orc.init_state_dict(ocfg, seed=5)
    inputs, targets, pos = orc.synthetic_batch(ocfg, n=2, t=2, h=6, w=5, seed=6)
    c = dict(kwargs=kw, state_dict=sd, inputs=inputs, targets=targets, pos_codes=pos)
    cfg, O_all, total, grads = run_engine(c)
    from hipvsr.plans import NetPlans
    assert NetPlans(cfg).xcol
    outs, loss, gref = orc.step(sd, ocfg, [x.clone() for x in inputs], targets, pos)
    torch.testing.assert_close(total, loss, atol=1e-5, rtol=1e-5)
    N = 2
    for g in range(3 * cfg.num_stages):
        for i in range(2):
            mine = O_all[g // 3, g % 3, i * N:(i + 1) * N].permute(0, 3, 1, 2)
            torch.testing.assert_close(mine, outs[g][i], atol=2e-5, rtol=1e-5)
    for k, gr in gref.items():
        if gr is None:
            assert grads[k] is None
            continue
        scale = float(gr.abs().max()) + 1e-12
        assert float((grads[k] - gr).abs().max()) <= 2e-4 * scale + 1e-7, k


def test_engine_last_group_only(g1):
    """Inference shortcut: only the fused group of the last stage goes through the upsampler."""
    c = g1['x4_pos1_mem1']
    cfg = NetConfig(**c['kwargs'])
    eng = RefineNetEngine(cfg, TorchOps('cpu'))
    params = {k: v.clone() for k, v in c['state_dict'].items()}
    full, _ = eng.forward(params, c['inputs'], c['pos_codes'], need_grad=False)
    last, _ = eng.forward(params, c['inputs'], c['pos_codes'], need_grad=False, last_only=True)
    assert torch.equal(last[-1, 2], full[-1, 2])


def test_winograd_geometry_choice_of_the_plans(monkeypatch):
    """Which geometry of rnh_conv_wino the plans ask for (hipvsr/plans.py ConvPlan.wino_cols): 128-column blocks (8 waves, one
    workgroup per CU) exactly where they cost no padding - column count a multiple of 128, every source a multiple of 32
    channels - and the matching ConvLSTM gate layout; 64-column blocks elsewhere; RNH_WINO_COLS=64 turns the wide geometry off;
    16-channel layers stay Winograd in the 64-column geometry; 8-channel layers fall back to the implicit GEMM and its layout."""
    from hipvsr.plans import NetPlans, lstm_colmap, lstm_colmap64
    from hipvsr.spec import NetConfig

    def cfg(nf):
        return NetConfig(1, 1, nf, num_stages=3, refine_window_size=5, upscale_factor=4, update_memory=True, num_updated_frames=6,
                         positional_encoding=True)
    P = NetPlans(cfg([64, 64, 64]))
    pl = P.lstm[('forward', 1)]
    assert pl['full'].wino and pl['full'].wino_cols == 128 and pl['full'].Npad == 256 and pl['full'].gate_group == 32
    assert pl['full'].colmap[:256] == lstm_colmap(64) and pl['first'].wino_cols == 128
    assert pl['dgrad'].wino and pl['dgrad'].wino_cols == 128 and pl['dgrad'].Npad == 128
    assert P.r1_wino and P.r1_fwd_h.wino_cols == 128 and P.r1_dgrad_h.wino_cols == 128
    assert P.r2_wino and P.r2_fwd_h.wino and P.r2_fwd_h.wino_cols == 64 and P.r2_dgrad_h.wino_cols == 128
    assert P.up[0]['fwd'].wino_cols == 128 and P.up[0]['dgrad'].wino and P.up[0]['dgrad'].wino_cols == 64   # 64 output columns
    monkeypatch.setenv('RNH_WINO_COLS', '64')
    P64 = NetPlans(cfg([64, 64, 64]))
    pl = P64.lstm[('forward', 1)]
    assert pl['full'].wino and pl['full'].wino_cols == 64 and pl['full'].gate_group == 16 and pl['full'].colmap[:256] == lstm_colmap64(64)
    assert P64.r1_fwd_h.wino_cols == 64 and P64.up[0]['fwd'].wino_cols == 64
    monkeypatch.delenv('RNH_WINO_COLS')
    P16 = NetPlans(cfg([16, 16]))
    pl = P16.lstm[('backward', 0)]
    assert pl['full'].wino and pl['full'].wino_cols == 64 and pl['full'].Npad == 64 and pl['full'].gate_group == 16
    P8 = NetPlans(cfg([8, 8]))
    pl = P8.lstm[('forward', 0)]
    assert not pl['full'].wino and pl['full'].gate_group == 32 and pl['full'].colmap[:len(lstm_colmap(8))] == lstm_colmap(8)


def test_bf16_last_channel_of_refine_conv1_frame_by_frame(monkeypatch):
    """bf16-storage path, round 3 (plans.xcol_m): the 129th output channel of refine conv1 computed frame by frame - one small
    convolution over the source frames against the view w1[2*Cl].view(w, C1, 3, 3), summed over the window slots - and its weight
    gradient written through the same view of the gradient, against (a) the 8-column launch it replaces (RNH_XCOL_M=0: same
    rounding points up to the order of two fp32 sums) and (b) the fp32 oracle (== reference refine_net.py:147-151, :166-183).
    Channel width 32 (the narrowest that takes this path), both through the torch double of the kernel interface."""
    from oracle import refinenet_oracle as orc
    from hipvsr.spec import state_dict_spec
    cfg_o = orc.Config(in_channels=1, out_channels=1, num_features=[32, 32], num_stages=2, refine_window_size=5, upscale_factor=4,
                       update_memory=True, num_updated_frames=2, positional_encoding=True)
    sd = orc.init_state_dict(cfg_o, seed=5)
    inputs, targets, pos = orc.synthetic_batch(cfg_o, n=2, t=2, h=6, w=5, seed=6)
    ref_out, ref_loss, ref_grads = orc.step(sd, cfg_o, [x.clone() for x in inputs], targets, pos)
    c = dict(kwargs=dict(cfg_o), state_dict=sd, inputs=inputs, targets=targets, pos_codes=pos)
    res = {}
    for flag in ('1', '0'):
        monkeypatch.setenv('RNH_XCOL_M', flag)
        cfg, O_all, total, grads = run_engine(c, dtype='bf16')
        res[flag] = (O_all, total, grads)
    from hipvsr.plans import NetPlans
    monkeypatch.setenv('RNH_XCOL_M', '1')
    P = NetPlans(NetConfig(**dict(cfg_o)), bf16=True)
    assert P.xcol_m and P.r1x_fwd.Npad == 64 and P.r1_wgrad_a.ycols_pad64 == 64 and P.r1x_key not in state_dict_spec(cfg)
    O1, t1, g1_ = res['1']
    O0, t0, g0_ = res['0']
    assert float((O1 - O0).norm()) <= 2e-3 * float(O0.norm())
    assert abs(float(t1) - float(t0)) <= 1e-3 * abs(float(t0))
    for k, v in g0_.items():
        if v is None:
            assert g1_[k] is None
        else:
            assert float((g1_[k] - v).norm()) <= 1e-2 * float(v.norm()) + 1e-7, (k, float((g1_[k] - v).norm()), float(v.norm()))
    assert abs(float(t1) - float(ref_loss)) <= 1e-2 * abs(float(ref_loss))
    k1 = 'refine_block.body.conv1.weight'
    last = g1_[k1][2 * 32]                                       # the row written through the view: all 5 x 65 x 9 entries
    assert float((last - ref_grads[k1][64]).norm()) <= 6e-2 * float(ref_grads[k1][64].norm())
    b1 = 'refine_block.body.conv1.bias'
    assert abs(float(g1_[b1][64]) - float(ref_grads[b1][64])) <= 6e-2 * float(ref_grads[b1].abs().max())


def test_full_width_refine_side_paths_vs_oracle():
    """Width 64 (the reference YAML's): refine conv1 takes the Winograd split with its side paths - channel 128 forward / weight gradient
    (xcol), the phase planes as a bias field forward and, since round 3, as border-class sums in the weight gradient
    (ops.refine_phase_wgrad) and the 45-tap stencil for channel 128's data gradient (ops.refine_xcol_dgrad).  The engine's index maps
    for all of them, through the torch double, against the oracle (== reference refine_net.py:147-151, :157-185): fp32 tolerances."""
    from oracle import refinenet_oracle as orc
    cfg_o = orc.Config(in_channels=1, out_channels=1, num_features=[64, 64], num_stages=2, refine_window_size=5, upscale_factor=2,
                       update_memory=True, num_updated_frames=2, positional_encoding=True)
    sd = orc.init_state_dict(cfg_o, seed=11)
    inputs, targets, pos = orc.synthetic_batch(cfg_o, n=2, t=2, h=5, w=6, seed=12)
    ref_out, ref_loss, ref_grads = orc.step(sd, cfg_o, [x.clone() for x in inputs], targets, pos)
    c = dict(kwargs=dict(cfg_o), state_dict=sd, inputs=inputs, targets=targets, pos_codes=pos)
    cfg, O_all, total, grads = run_engine(c)
    from hipvsr.plans import NetPlans
    P = NetPlans(cfg)
    assert P.r1_wino and P.xcol and P.r1_cols == 128
    torch.testing.assert_close(total, ref_loss, atol=1e-5, rtol=1e-5)
    for k, gref in ref_grads.items():
        if gref is None:
            assert grads[k] is None
            continue
        scale = float(gref.abs().max()) + 1e-12
        err = float((grads[k] - gref).abs().max())
        assert err <= 2e-4 * scale + 1e-7, (k, err, scale)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('case', ['x4_pos1_mem1', 'x2_pos0_mem1'])
def test_fused_gate_backward_schedule_equals_the_two_launch_schedule(g1, monkeypatch, case, dtype):
    """The skewed wavefront with the gate backward in the data-gradient launch's epilogue (engine.backward, conv(..., lstm_bwd=...))
    asks for the same arithmetic as a gate-backward launch + a data-gradient launch per cell and frame: on the test double every
    gradient comes out bit-identical, and both match the reference's goldens (the first parametrised tests of this file)."""
    calls = {'fused': 0, 'gates': 0}
    conv, gates = TorchOps.conv, TorchOps.lstm_gates_bwd

    def count_conv(self, *a, **k):
        calls['fused'] += k.get('lstm_bwd') is not None
        return conv(self, *a, **k)

    def count_gates(self, *a, **k):
        calls['gates'] += 1
        return gates(self, *a, **k)

    monkeypatch.setattr(TorchOps, 'conv', count_conv)
    monkeypatch.setattr(TorchOps, 'lstm_gates_bwd', count_gates)
    monkeypatch.setenv('RNH_FUSE_GATES_BWD', '0')
    _, _, tot_a, ga = run_engine(g1[case], dtype=dtype)
    assert calls['fused'] == 0 and calls['gates'] > 0
    two = calls['gates']
    calls['gates'] = 0
    monkeypatch.setenv('RNH_FUSE_GATES_BWD', '1')
    monkeypatch.setenv('RNH_FUSE_ANY', '1')
    cfg, _, tot_b, gb = run_engine(g1[case], dtype=dtype)
    T = len(g1[case]['targets'])
    chains = cfg.num_stages * 2 * len(cfg.num_features)
    assert calls['fused'] == chains * (T - 1)                    # every frame but the head of its chain
    assert calls['gates'] == two                                 # (the double's fused op calls its own gate backward: same count)
    assert torch.equal(tot_a, tot_b)
    for k in ga:
        assert (ga[k] is None) == (gb[k] is None)
        if ga[k] is not None:
            assert torch.equal(ga[k], gb[k]), k


# ---------------------------------------------------------------------------------------------------------------------
# round 4: activation-memory plan (liveness of the ConvLSTM states, gate recomputation)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('mode', ['recompute', '1'])
@pytest.mark.parametrize('fuse', ['0', 'any'])
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('case', ['x4_pos1_mem1', 'x2_pos1_mem0', 'x3_pos0_mem1'])
def test_gate_recomputation_equals_the_stored_gates_step_bitwise(g1, monkeypatch, case, dtype, fuse, mode):
    """RNH_GATES=recompute: the forward stores no gates, the backward re-runs the cell's forward launch in front of every gate
    backward (engine.backward, gates_of) - same operands, same operator, so loss and every gradient must equal the stored-gates step
    bit for bit, in the frame-by-frame and in the fused (skewed) BPTT schedule; the extra launches are counted.  mode '1': only the
    first stage recomputes (what 'auto' picks when all but one stage's gates fit)."""
    calls = {'cell': 0}
    conv = TorchOps.conv

    def count_conv(self, plan, *a, **k):
        calls['cell'] += k.get('lstm') is not None
        return conv(self, plan, *a, **k)

    monkeypatch.setattr(TorchOps, 'conv', count_conv)
    monkeypatch.setenv('RNH_FUSE_GATES_BWD', '0' if fuse == '0' else '1')
    if fuse == 'any':
        monkeypatch.setenv('RNH_FUSE_ANY', '1')
    monkeypatch.setenv('RNH_GATES', 'store')
    cfg, Oa, ta, ga = run_engine(g1[case], dtype=dtype)
    n_store = calls['cell']
    calls['cell'] = 0
    monkeypatch.setenv('RNH_GATES', mode)
    _, Ob, tb, gb = run_engine(g1[case], dtype=dtype)
    T = len(g1[case]['targets'])
    stages = cfg.num_stages if mode == 'recompute' else 1
    assert calls['cell'] == n_store + stages * 2 * len(cfg.num_features) * T     # one more cell launch per (cell, supervised frame) of those stages
    assert torch.equal(Oa, Ob) and torch.equal(ta, tb)
    for k in ga:
        assert (ga[k] is None) == (gb[k] is None)
        if ga[k] is not None:
            assert torch.equal(ga[k], gb[k]), k


def test_forward_keeps_only_what_the_backward_reads(g1, monkeypatch):
    """Liveness: after the forward a stage holds, per direction, the hidden states of the lower layers and the cell
    states of T + 1 frames (the supervised ones and the one in front of them), the top layer's hidden states of the frames the stage
    computed, the gates / features / refine intermediates of the T supervised frames - not F frames of everything; frames outside are
    gone (an access raises), and the bytes agree with engine.memory_plan()['per_stage'] to the byte."""
    from hipvsr.engine import FrameStore
    monkeypatch.setenv('RNH_GATES', 'store')
    c = g1['x4_pos1_mem1']
    cfg = NetConfig(**c['kwargs'])
    eng = RefineNetEngine(cfg, TorchOps('cpu'))
    params = {k: v.clone() for k, v in c['state_dict'].items()}
    _, ctx = eng.forward(params, c['inputs'], c['pos_codes'], need_grad=True)
    N, _, H, W = c['inputs'][0].shape
    F, U, S = len(c['inputs']), cfg.num_updated_frames, cfg.num_stages
    T, hw, Lr = F - 2 * U, cfg.refine_window_size // 2, len(cfg.num_features)
    plan = eng.memory_plan(N, H, W, F)
    nb = lambda t: t.numel() * t.element_size()                       # noqa: E731
    kept = 0
    for s, st in enumerate(ctx.stages):
        per = dict(h_lower=0, c=0, gates=0, feat=st['feat'].nbytes(), r1=nb(st['R1']), sb=nb(st['Sb']), ys=sum(nb(y) for y in st['Ys']))
        top = 0
        for d, (klo, khi) in (('forward', (U - 1, U + T)), ('backward', (U, U + T + 1))):
            for l in range(Lr):
                hs, cs = st[d]['H'][l], st[d]['C'][l]
                assert isinstance(hs, FrameStore) and cs.ring is None and (cs.klo, cs.khi) == (klo, khi)
                per['c'] += cs.nbytes()
                if l < Lr - 1:
                    assert (hs.lo, hs.hi) == (klo, khi)
                    per['h_lower'] += hs.nbytes()
                    with pytest.raises(IndexError):
                        hs.view(klo - 1)
                    with pytest.raises(IndexError):
                        hs.view(khi)
                else:
                    top += hs.nbytes()
                with pytest.raises(IndexError):
                    cs.view(khi)
                per['gates'] += nb(st[d]['G'][l])
        assert (st['feat'].lo, st['feat'].hi) == (U, U + T) and st['R1'].shape[0] == T * N
        frames_top = F if s < S - 1 else U + T + hw
        assert top == 2 * frames_top * N * H * W * cfg.num_features[-1] * 4
        assert per == plan['per_stage'], (s, per, plan['per_stage'])
        kept += sum(per.values()) + top
    assert kept == plan['kept']
    # and the same step with recomputation keeps no gates at all
    monkeypatch.setenv('RNH_GATES', 'recompute')
    _, ctx2 = eng.forward(params, c['inputs'], c['pos_codes'], need_grad=True)
    assert ctx2.recompute and all(st[d]['G'] is None for st in ctx2.stages for d in ('forward', 'backward'))
    assert eng.memory_plan(N, H, W, F, recompute=True)['per_stage']['gates'] == 0


def test_cell_state_ring_and_frame_store_pieces():
    """FrameStore: three separately allocated pieces around ``keep``; ring mode shares two slots in processing order, in both directions."""
    from hipvsr.engine import FrameStore
    ops = TorchOps('cpu')
    fs = FrameStore(ops, 2, 0, 10, (3, 6), (4, 4, 8), torch.float32)
    assert [(a, b, k) for a, b, _, k in fs.segs] == [(0, 3, False), (3, 6, True), (6, 10, False)]
    assert fs.view(4).data_ptr() == fs.segs[1][2][2:4].data_ptr() and fs.span(3, 3)[1] == 0 and fs.pieces(2, 7) == [(2, 3), (3, 6), (6, 7)]
    with pytest.raises(IndexError):
        fs.span(2, 3)                                               # straddles two pieces
    fs.release()
    assert len(fs.segs) == 1 and fs.frames(3, 6).shape[0] == 6
    for step, lo, hi in ((1, 0, 9), (-1, 2, 11)):
        rs = FrameStore(ops, 2, lo, hi, (4, 7), (4, 4, 8), torch.float32, ring=2, step=step)
        order = list(range(lo, hi)) if step > 0 else list(range(hi - 1, lo - 1, -1))
        prev = None
        for k in order:
            t, o = rs.loc(k)
            if not 4 <= k < 7:
                assert t is rs.ring
                if prev is not None and prev[0] is rs.ring:
                    assert o != prev[1]                              # a cell never writes the slot it reads its predecessor from
            else:
                assert t is not rs.ring
            prev = (t, o)
    none = FrameStore(ops, 2, 0, 5, None, (4, 4, 8), torch.float32, ring=2)
    assert none.segs == [] and none.ring.shape[0] == 4


def test_auto_gate_plan_recomputes_only_where_the_stored_step_does_not_fit():
    """'auto' (the default): BASELINE config 2 / 5 and config 4 in bf16 store their gates on a 288 GB card, config 4 in fp32 (N = 16, T = 5,
    256 x 256: an estimated 235 GB peak with stored gates, 154 GB without) recomputes them; RNH_GATES / gate_memory override."""
    from oracle import refinenet_oracle as orc

    class Ops288(TorchOps):
        def total_memory(self):
            return 288 * 10**9
    for over, n, t, size, dt, want in ((dict(), 8, 7, 128, 'f32', False), (dict(), 8, 11, 96, 'f32', False),
                                       (dict(upscale_factor=2), 16, 5, 256, 'bf16', False), (dict(upscale_factor=2), 16, 5, 256, 'f32', True),
                                       (dict(upscale_factor=2), 4, 5, 256, 'f32', False)):
        eng = RefineNetEngine(NetConfig(**orc.exp1_x4_config(**over)), Ops288('cpu'), dtype=dt)
        assert eng.recompute_gates(n, size, size, t + 12) is want, (over, n, dt)
        eng.gate_memory = 'recompute'
        assert eng.recompute_gates(n, size, size, t + 12) is True
    assert RefineNetEngine(NetConfig(**orc.exp1_x4_config()), TorchOps('cpu')).recompute_gates(64, 512, 512, 19) is False    # no device: store


def test_default_forms_at_the_benchmark_shapes(monkeypatch):
    """VERDICT r05 item 6 / ADVICE r05: ONE record of the forms a step runs in (hipvsr/forms.py, RefineNetEngine.resolve_forms) - what the forward, the
    backward, memory_plan, the weight packing and bench.py's `config.forms` all read.  Its defaults at the benchmark shapes on a 288 GB card are what
    DESIGN section 4 says: fp32 cells in F(4x4, 3x3) form everywhere (whole 4x4 tiles), a transformed-h' slot per frame at configs 2, 5 and the YAML
    shape, a ring of four at config 4 (62 GB of slots) - where a CAPTURED step falls back to F(2x2) cells and the record says so -, refine conv1 and
    the first PixelShuffle convolution following the cells, the cell's data gradient in F(4x4) form on the transformed gate gradients the gate backward
    writes (round 6; where the tiles come in whole 8 x 4 blocks - every benchmark shape); bf16: direct forms, IEEE-half weights in the
    upsampler's forward.  An environment switch set to anything but the product's choice is listed."""
    from hipvsr import forms
    from oracle import refinenet_oracle as orc
    for k in forms.SWITCHES:
        monkeypatch.delenv(k, raising=False)

    class HipLike(TorchOps):                      # the double with the attributes resolve_forms asks a HIP backend for
        wino44_cell = None

        def total_memory(self):
            return 288 * 10**9

        def pair_cells(self, N, H, W):
            return True

        def wino44_gates_bwd_supported(self, H, W, hd):          # (rnh_wino44_gates_bwd_supported)
            return H % 16 == 0 and W % 32 == 0 and hd % 16 == 0

        def _wgrad44f_v(self, *a, **k):                          # (HipOps has the transformed-images form of the fused weight gradient)
            return False
    cases = {'config 2': (dict(), 8, 7, 128), 'config 4': (dict(upscale_factor=2), 16, 5, 256), 'config 5': (dict(), 8, 11, 96), 'yaml': (dict(), 16, 7, 32)}
    got = {}
    for name, (over, n, t, size) in cases.items():
        eng = RefineNetEngine(NetConfig(**orc.exp1_x4_config(**over)), HipLike('cpu'))
        fm = got[name] = eng.resolve_forms(n, size, size, t + 12)
        assert fm.cells44 and not fm.capture_fallback and fm.cell_dgrad44 and fm.gates_bwd44 and fm.paired and fm.refine_dgrad44, name
        assert all(fm.uses44(eng.plans.lstm[k][kind]) for k in eng.plans.lstm for kind in ('full', 'first'))
        assert all(fm.uses44(eng.plans.lstm[k]['dgrad']) for k in eng.plans.lstm)
        d = fm.describe()
        assert 'F(4x4,3x3)' in d['cell'] and 'F(4x4,3x3)' in d['cell_dgrad'] and 'rnh_wino44_gates_bwd' in d['cell_dgrad'] and d['env_overrides'] == [] and d['paired'] is True
        assert 'rnh_wino44f_wgrad' in d['cell_wgrad'] and 'rnh_wino44f_wgrad' in d['refine1_wgrad']
        # memory_plan reads the same record
        assert eng.memory_plan(n, size, size, t + 12)['forward_transient'] > RefineNetEngine(NetConfig(**orc.exp1_x4_config(**over)), TorchOps('cpu')).memory_plan(n, size, size, t + 12)['forward_transient']
    for name in ('config 2', 'config 5', 'yaml'):
        fm = got[name]
        assert fm.ring == 0 and fm.refine_fwd44 and fm.up44 == [True] and fm.recompute == 0, name
        assert "slot per frame" in fm.describe()['cell'] and 'F(4x4,3x3)' in fm.describe()['refine1_fwd'] and 'F(4x4,3x3)' in fm.describe()['up1_fwd']
    # round 6: where every frame's transformed image has a slot and all stages' images take at most 12 % of the card, the weight gradients copy their x operand
    # from them (rnh_wino44f_wgrad_v) - and memory_plan counts what is kept; a ring shape, a small card or the switch: the raw-operand form
    for name in ('config 2', 'config 5', 'yaml'):
        fm = got[name]
        assert fm.wgrad_v and fm.refine1_wgrad_v and 'rnh_wino44f_wgrad_v' in fm.describe()['cell_wgrad'] and 'rnh_wino44f_wgrad_v' in fm.describe()['refine1_wgrad'], name
        assert not fm.refine2_fwd44 and not fm.refine2_dgrad44, name                         # (refine conv2 in F(4x4) form is an opt-in)
    assert not got['config 4'].wgrad_v and 'rnh_wino44f_wgrad_v' not in got['config 4'].describe()['cell_wgrad']

    class SmallCard(HipLike):
        def total_memory(self):
            return 150 * 10**9
    sm = RefineNetEngine(NetConfig(**orc.exp1_x4_config()), SmallCard('cpu')).resolve_forms(8, 128, 128, 19)
    assert sm.cells44 and not sm.ring and not sm.wgrad_v                                     # (30 GB of kept images > 12 % of 150 GB; the slots of one stage, 8.6 GB, are under its 8 %)
    monkeypatch.setenv('RNH_WINO44F_V', '0')
    off = RefineNetEngine(NetConfig(**orc.exp1_x4_config()), HipLike('cpu'))
    assert not off.resolve_forms(8, 128, 128, 19).wgrad_v and off.resolve_forms(8, 128, 128, 19).describe()['env_overrides'] == ['RNH_WINO44F_V=0']
    kept_raw = off.memory_plan(8, 128, 128, 19)['kept']
    monkeypatch.delenv('RNH_WINO44F_V')
    kept_v = RefineNetEngine(NetConfig(**orc.exp1_x4_config()), HipLike('cpu')).memory_plan(8, 128, 128, 19)['kept']
    assert kept_v - kept_raw == 3 * ((19 * 64 + 2 * 19 * 192) + 3 * 7 * 64) * 8 * 128 * 128 * 9          # features + h' of every frame, the PixelShuffle conv's input: 34.9 GB (an upper bound: the last stage holds fewer frames)
    fm4 = got['config 4']
    assert fm4.ring == 4 and not fm4.refine_fwd44 and fm4.up44 == [] and fm4.recompute >= 1 and 'ring of 4' in fm4.describe()['cell']
    assert 'F(2x2,3x3)' in fm4.describe()['refine1_fwd'] and 'recomputed in' in fm4.describe()['gates']
    # a captured step at the ring shape: F(2x2) cells, said so; at a slot shape capture changes nothing
    eng4 = RefineNetEngine(NetConfig(**orc.exp1_x4_config(upscale_factor=2)), HipLike('cpu'))
    cap = eng4.resolve_forms(16, 256, 256, 17, capturing=True)
    assert cap.capture_fallback and not cap.cells44 and not cap.plans44 and 'capture fallback' in cap.describe()['cell'] and cap.describe()['graph_capture']
    eng2 = RefineNetEngine(NetConfig(**orc.exp1_x4_config()), HipLike('cpu'))
    assert eng2.resolve_forms(8, 128, 128, 19, capturing=True).cells44
    # images that are not whole 4x4 tiles: every launch in its F(2x2) form; whole tiles but not whole 8 x 4 blocks of them (the fused gate backward's
    # geometry): the cells in F(4x4) form, their data gradient in F(2x2) form
    odd = eng2.resolve_forms(2, 33, 20, 15)
    assert not odd.cells44 and not odd.plans44 and 'F(2x2,3x3)' in odd.describe()['cell']
    t44 = eng2.resolve_forms(2, 24, 40, 15)
    assert t44.cells44 and not t44.cell_dgrad44 and not t44.gates_bwd44 and 'F(2x2,3x3)' in t44.describe()['cell_dgrad']
    # bf16 storage: direct forms; the upsampler's forward with IEEE-half weights
    engb = RefineNetEngine(NetConfig(**orc.exp1_x4_config()), HipLike('cpu'), dtype='bf16')
    fb = engb.resolve_forms(8, 128, 128, 19)
    db = fb.describe()
    assert not fb.cells44 and not fb.plans44 and 'bf16 MFMA' in db['cell'] and 'IEEE-half' in db['up1_fwd'] and 'f16 MFMA' in db['tail']
    assert engb.plans.up[0]['fwd'].f16w and not getattr(engb.plans.up[0]['dgrad'], 'f16w', False)
    # an A/B switch shows up in the record (and makes bench.py refuse the line without --allow-overrides)
    monkeypatch.setenv('RNH_WINO44', '0')
    eng2b = RefineNetEngine(NetConfig(**orc.exp1_x4_config()), HipLike('cpu'))
    off = eng2b.resolve_forms(8, 128, 128, 19)
    assert not off.cells44 and off.describe()['env_overrides'] == ['RNH_WINO44=0']
    monkeypatch.setenv('RNH_WINO44', '1')                     # (the product's own value: not an override)
    assert forms.env_overrides() == []


def test_input_block_width_outside_the_backward_kernels_set_plans_for_inference_only():
    """num_features[0] = 24: the forward kernels serve it, rnh_inconv_prelu_bwd does not (4, 8, ..., 256).  The net is planned and runs
    forward without gradients (a predict-only user must not be refused at construction); the first forward that is asked to
    keep what a backward needs raises ValueError with the reason."""
    from oracle import refinenet_oracle as orc
    kw = dict(in_channels=1, out_channels=1, num_features=[24, 24], num_stages=2, refine_window_size=5, upscale_factor=2,
              update_memory=True, num_updated_frames=2, positional_encoding=True)
    cfg = NetConfig(**kw)
    eng = RefineNetEngine(cfg, TorchOps('cpu'))
    assert eng.plans.inconv_bwd_error and '24' in eng.plans.inconv_bwd_error
    ocfg = orc.Config(**kw)
    sd = orc.init_state_dict(ocfg, seed=3)
    inputs, targets, pos = orc.synthetic_batch(ocfg, n=1, t=2, h=6, w=5, seed=4)
    O, ctx = eng.forward({k: v.clone() for k, v in sd.items()}, inputs, pos, need_grad=False)
    assert ctx is None
    with torch.no_grad():
        ref = orc.forward(orc.as_leaf_params(sd), ocfg, inputs, pos)
    mine = O[cfg.num_stages - 1, 2, 0:1].permute(0, 3, 1, 2)
    torch.testing.assert_close(mine, ref[-1][0], atol=2e-5, rtol=1e-5)
    with pytest.raises(ValueError, match='backward'):
        eng.forward({k: v.clone() for k, v in sd.items()}, inputs, pos, need_grad=True)
    assert RefineNetEngine(NetConfig(**dict(kw, num_features=[32, 32])), TorchOps('cpu')).plans.inconv_bwd_error is None


def test_an_exception_inside_the_engine_drains_the_side_streams_before_it_propagates(g1):
    """forward / backward raise between aside() and rejoin() -> the buffers of the unwinding stack go back to the allocator while
    the helper stream may still use them.  The engine calls ops.quiesce() (HipOps: device synchronise + helper bookkeeping reset) before the
    exception leaves it - in the forward and in the backward."""
    c = g1['x4_pos1_mem1']
    cfg = NetConfig(**c['kwargs'])

    class Boom(RuntimeError):
        pass

    class Ops(TorchOps):
        quiesced, fail_at, calls = 0, -1, 0

        def quiesce(self):
            self.quiesced += 1

        def conv(self, *a, **k):
            self.calls += 1
            if self.calls == self.fail_at:
                raise Boom('launch refused')
            return super().conv(*a, **k)

    ops = Ops('cpu')
    eng = RefineNetEngine(cfg, ops)
    params = {k: v.clone() for k, v in c['state_dict'].items()}
    ops.fail_at = 7
    with pytest.raises(Boom):
        eng.forward(params, c['inputs'], c['pos_codes'], need_grad=True)
    assert ops.quiesced == 1
    ops.fail_at, ops.calls = -1, 0
    O, ctx = eng.forward(params, c['inputs'], c['pos_codes'], need_grad=True)        # the engine is usable afterwards
    n_fwd = ops.calls
    ops.fail_at = n_fwd + 3
    with pytest.raises(Boom):
        eng.backward(params, ctx, torch.ones_like(O) * 1e-3)
    assert ops.quiesced == 2


@pytest.mark.parametrize('case,dtype', [('x4_pos1_mem1', 'f32'), ('x2_pos1_mem0', 'f32'), ('x4_pos1_mem1', 'bf16'), ('x3_pos0_mem1', 'bf16')])
def test_engine_with_paired_direction_launches_matches_reference_golden(g1, case, dtype, monkeypatch):
    """The product pairs the two directions' ConvLSTM cells (and data gradients) of a layer into one launch on one stream per layer
    (ops.conv_pair, rnh_conv_*_pair).  The double executes a pair as its two calls in turn: what is checked here is the engine's paired wavefront -
    forward, the fused and the unfused back-propagation through time - against the reference's outputs and gradients, and against the unpaired run."""
    c = g1[case]
    monkeypatch.setenv('RNH_PAIR', '0')
    _, O0, t0, g0 = run_engine(c, dtype=dtype)
    monkeypatch.setenv('RNH_PAIR', '1')
    cfg, O1, t1, grads = run_engine(c, dtype=dtype)
    assert torch.equal(O0, O1) and torch.equal(t0, t1)
    for k, v in g0.items():
        assert (v is None and grads[k] is None) or torch.equal(v, grads[k]), k
    if dtype == 'f32':
        torch.testing.assert_close(t1, c['train_loss'], atol=1e-5, rtol=1e-5)
        for k, gref in c['grads'].items():
            if gref is not None:
                assert float((grads[k] - gref).abs().max()) <= 2e-4 * float(gref.abs().max()) + 1e-7, k
