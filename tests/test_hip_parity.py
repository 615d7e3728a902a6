"""Parity of the HIP path (through the C ABI) with the reference, on the GPU box.

* golden vectors captured from the reference itself (tests/golden/g1_tiny.pt, g2_cfg1.pt, g5_edges.pt);
* the oracle (oracle/refinenet_oracle.py) on seeded inputs at sizes it finishes in seconds;
* kernel-level comparisons against the torch semantics in tests/torch_ops.py on shapes that exercise tile
  tails, several M tiles, every epilogue and every tile shape.

Tolerances (fp32, K up to 5 805 re-associated products): outputs atol 1e-4 / rtol 1e-4; gradients
|delta| <= 1e-3 * max|g| (+1e-6); loss rtol 1e-5; PSNR |delta| < 0.01 dB (north_star).
"""
import os

import numpy as np
import pytest
import torch

from oracle import refinenet_oracle as orc

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def g1(golden_dir):
    return torch.load(os.path.join(golden_dir, 'g1_tiny.pt'), weights_only=False)


CASES = [f'x{s}_pos{p}_mem{m}' for s in (2, 3, 4) for p in (1, 0) for m in (1, 0)] + ['x8_pos1_mem1']


def _grad_close(mine, ref, name, rel=1e-3):
    scale = float(ref.abs().max())
    err = float((mine.detach().cpu() - ref).abs().max())
    assert err <= rel * scale + 1e-6, (name, err, scale)


def _module_step(kwargs, sd, inputs, targets, pos, loss_fn):
    from src.model.nets import RefineNet
    from src.runner.trainers import AcdcVSRRefineNetTrainer
    dev = _dev()
    net = RefineNet(**kwargs)
    net.load_state_dict(sd)
    net = net.to(dev)
    tr = object.__new__(AcdcVSRRefineNetTrainer)
    tr.net, tr.loss_fns, tr.metric_fns = net, [loss_fn], []
    net.train()
    outs = net([x.to(dev) for x in inputs], pos.to(dev))
    loss = tr._compute_losses(outs, [t.to(dev) for t in targets])[0]
    net.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    return net, tr, outs, loss


@pytest.mark.parametrize('case', CASES)
def test_module_forward_backward_vs_reference_golden(g1, case):
    c = g1[case]
    net, tr, outs, loss = _module_step(c['kwargs'], c['state_dict'], c['inputs'], c['targets'], c['pos_codes'],
                                       torch.nn.L1Loss())
    assert list(net.state_dict().keys()) == list(c['state_dict'].keys())
    assert len(outs) == len(c['outputs'])
    for go, gr in zip(outs, c['outputs']):
        for a, b in zip(go, gr):
            assert a.shape == b.shape
            torch.testing.assert_close(a.detach().cpu(), b, atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(loss.detach().cpu(), c['train_loss'], atol=1e-6, rtol=1e-5)
    for k, p in net.named_parameters():
        if c['grads'][k] is None:
            assert p.grad is None, k                    # quirk Q1
        else:
            _grad_close(p.grad, c['grads'][k], k)
    # evaluation branch of the loss schedule
    net.eval()
    dev = _dev()
    with torch.no_grad():
        outs_e = net([x.to(dev) for x in c['inputs']], c['pos_codes'].to(dev))
        le = tr._compute_losses(outs_e, [t.to(dev) for t in c['targets']])[0]
    torch.testing.assert_close(le.cpu(), c['eval_loss'], atol=1e-6, rtol=1e-5)
    for a, b in zip(outs_e[-1], c['eval_last']):
        torch.testing.assert_close(a.cpu(), b, atol=1e-4, rtol=1e-4)


def test_module_charbonnier_fused_loss(g1):
    from src.model.losses import CharbonnierLoss
    c = g1['x4_pos1_mem1']
    net, _, _, loss = _module_step(c['kwargs'], c['state_dict'], c['inputs'], c['targets'], c['pos_codes'],
                                   CharbonnierLoss(epsilon=1e-6))
    torch.testing.assert_close(loss.detach().cpu(), c['charbonnier_train_loss'], atol=1e-6, rtol=1e-5)
    for k, p in net.named_parameters():
        if c['charbonnier_grads'][k] is not None:
            _grad_close(p.grad, c['charbonnier_grads'][k], k)


def test_unfused_loss_path_matches(g1):
    """A loss the fused kernel does not serve (Huber) goes through loss_fn per pair and plain autograd views."""
    from src.model.losses import HuberLoss
    c = g1['x2_pos1_mem1']
    net, _, outs, loss = _module_step(c['kwargs'], c['state_dict'], c['inputs'], c['targets'], c['pos_codes'],
                                      HuberLoss(delta=0.5))
    cfg = orc.Config(**c['kwargs'])
    _, lref, gref = orc.step(c['state_dict'], cfg, [x.clone() for x in c['inputs']], c['targets'], c['pos_codes'],
                             loss_fn=lambda o, t: orc.huber_loss(o, t, 0.5))
    torch.testing.assert_close(loss.detach().cpu(), lref, atol=1e-6, rtol=1e-5)
    for k, p in net.named_parameters():
        if gref[k] is not None:
            _grad_close(p.grad, gref[k], k)


def test_whole_cycle_inference_golden(golden_dir):
    """predictor-style evaluation: batch 1, F = 30 + 12 frames, odd 7x9 frames (g5)."""
    from src.model.nets import RefineNet
    r = torch.load(os.path.join(golden_dir, 'g5_edges.pt'), weights_only=False)['cycle']
    dev = _dev()
    net = RefineNet(**r['kwargs'])
    net.load_state_dict(r['state_dict'])
    net = net.to(dev).eval()
    with torch.no_grad():
        last = net([x.to(dev) for x in r['inputs']], r['pos_codes'].to(dev))[-1]
    assert len(last) == 30
    for a, b in zip(last, r['last']):
        torch.testing.assert_close(a.cpu(), b, atol=1e-4, rtol=1e-4)


def test_cfg1_full_width_vs_reference_digest(golden_dir):
    """BASELINE config 1: x4, N=1, T=3, 64x64 -> 256x256, num_features [64,64,64]; digests from the reference."""
    r = torch.load(os.path.join(golden_dir, 'g2_cfg1.pt'), weights_only=False)
    cfg = orc.exp1_x4_config()
    sd = orc.init_state_dict(cfg, seed=r['seed_weights'])
    inputs, targets, pos = orc.synthetic_batch(cfg, r['n'], r['t'], r['h'], r['w'], seed=r['seed_inputs'])
    from src.model.metrics import PSNR
    net, tr, outs, loss = _module_step(dict(cfg), sd, inputs, targets, pos, torch.nn.L1Loss())
    assert abs(float(loss.detach()) - r['train_loss']) <= 1e-5 * abs(r['train_loss'])
    for g, grp in enumerate(outs):
        for i, o in enumerate(grp):
            torch.testing.assert_close(o.detach().cpu()[0, 0, 100:116, 100:116], r['out_crop'][g][i], atol=1e-4, rtol=1e-4)
            assert abs(float(o.double().sum()) - r['out_sum'][g][i]) <= 1e-4 * r['out_abs_sum'][g][i]
    for k, p in net.named_parameters():
        if r['grad_l2'][k] is None:
            assert p.grad is None
        else:
            assert abs(float(p.grad.double().norm()) - r['grad_l2'][k]) <= 1e-3 * r['grad_l2'][k], k
            _grad_close(p.grad.flatten()[:16], r['grad_head'][k], k, rel=1e-2)
    import functools
    from src.utils import denormalize
    tr.metric_fns = [PSNR().to(_dev())]
    tr._denormalize = functools.partial(denormalize, dataset='acdc')
    psnr = float(tr._compute_metrics(outs, [t.to(_dev()) for t in targets])[0])
    assert abs(psnr - r['psnr']) < 0.01                                  # north_star: |delta PSNR| < 0.01 dB


@pytest.mark.parametrize('direct', [True, False])
def test_kernels_vs_torch_semantics_on_ragged_shapes(direct):
    """Both variants of rnh_conv_igemm (DIRECT: fragments straight from global memory; LDS-staged).
    Full-width channels (64) at 20x13 (no dimension a multiple of any tile), N=2, T=2: the HIP engine against the
    same engine over the torch double - every kernel, every epilogue, M tails, multi-tile grids."""
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
    for name, ops, d in (('hip', HipOps(dev, direct=direct), dev), ('ref', TorchOps('cpu'), torch.device('cpu'))):
        eng = RefineNetEngine(cfg, ops)
        params = {k: v.to(d) for k, v in sd.items()}
        O, ctx = eng.forward(params, [x.to(d) for x in inputs], pos.to(d), need_grad=True)
        g = torch.Generator('cpu').manual_seed(9)
        dO = (torch.randn(O.shape, generator=g) * 1e-3).to(d)
        grads = eng.backward(params, ctx, dO)
        res[name] = (O.cpu(), {k: (v.cpu() if v is not None else None) for k, v in grads.items()})
    torch.testing.assert_close(res['hip'][0], res['ref'][0], atol=1e-4, rtol=1e-4)
    for k, v in res['ref'][1].items():
        if v is not None:
            _grad_close(res['hip'][1][k], v, k)


@pytest.mark.parametrize('r,B,Hm,Wm,C1,Cq', [(2, 2, 1, 1, 64, 64), (2, 1, 3, 2, 64, 64), (2, 2, 19, 37, 64, 64), (3, 1, 2, 5, 64, 64),
                                             (3, 2, 17, 33, 64, 64), (2, 1, 40, 16, 32, 16), (3, 1, 16, 48, 16, 32)])
def test_fused_tail_forward_vs_torch(r, B, Hm, Wm, C1, Cq):
    """rnh_uptail_fwd (last PixelShuffle conv + final conv as one composed 5x5 convolution, exact path sums in the
    border band) against conv2d -> pixel_shuffle -> conv2d, including images smaller than the band and tile tails."""
    import torch.nn.functional as F
    from hipvsr.hip_ops import HipOps
    dev = _dev()
    ops = HipOps(dev)
    g = torch.Generator('cpu').manual_seed(100 * r + Hm)
    y1 = torch.randn(B, Hm, Wm, C1, generator=g)
    w2 = torch.randn(Cq * r * r, C1, 3, 3, generator=g) * 0.05
    b2 = torch.randn(Cq * r * r, generator=g) * 0.1
    w3 = torch.randn(1, Cq, 3, 3, generator=g) * 0.05
    b3 = torch.randn(1, generator=g)
    ref = F.conv2d(F.pixel_shuffle(F.conv2d(y1.permute(0, 3, 1, 2).double(), w2.double(), b2.double(), padding=1), r),
                   w3.double(), b3.double(), padding=1).permute(0, 2, 3, 1)
    out = torch.full((B, Hm * r, Wm * r, 1), float('nan'), device=dev)
    ops.uptail_fwd(y1.to(dev), w2.to(dev), b2.to(dev), w3.to(dev), b3.to(dev), r, out)
    torch.cuda.synchronize()
    torch.testing.assert_close(out.cpu().double(), ref, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize('r,B,Hm,Wm,C1', [(2, 2, 1, 1, 64), (2, 1, 1, 7, 64), (2, 1, 5, 1, 16), (2, 3, 19, 37, 64), (3, 2, 17, 35, 64),
                                          (2, 1, 33, 16, 128), (3, 1, 9, 40, 32), (4, 1, 6, 9, 16), (2, 2, 8, 6, 8)])
def test_collapsed_tail_backward_kernels_vs_torch(r, B, Hm, Wm, C1):
    """rnh_uptail_compose + rnh_uptail_dgrad (merged-offset tile kernel + border term where C1 % 16 == 0, generic kernel
    otherwise) and rnh_uptail_xcorr (M, S on the matrix cores + border term) against the layer-by-layer torch backward,
    on single-row / single-column images, tile tails and several persistent blocks."""
    from hipvsr.hip_ops import HipOps
    from torch_ops import TorchOps
    dev = _dev()
    ops, ref = HipOps(dev), TorchOps('cpu')
    g = torch.Generator('cpu').manual_seed(7 * r + Hm + C1)
    Cq = 8
    y1 = torch.randn(B, Hm, Wm, C1, generator=g)
    d_o = torch.randn(B, Hm * r, Wm * r, 1, generator=g)
    w2 = torch.randn(Cq * r * r, C1, 3, 3, generator=g) * 0.1
    w3 = torch.randn(1, Cq, 3, 3, generator=g) * 0.1
    G = ops.uptail_compose(w2.to(dev), w3.to(dev), r)
    dy1 = ops.uptail_dgrad(d_o.to(dev), G, C1, r)
    torch.cuda.synchronize()
    want = ref.uptail_dgrad(d_o, ref.uptail_compose(w2, w3, r), C1, r)
    _grad_close(dy1, want, 'dY1', rel=2e-5)
    if ops.uptail_xcorr_supported(C1, r, 1):
        M, S = ops.uptail_xcorr(y1.to(dev), d_o.to(dev), r)
        torch.cuda.synchronize()
        Mr, Sr = ref.uptail_xcorr(y1, d_o, r)
        _grad_close(M, Mr, "M", rel=2e-5)
        _grad_close(S, Sr, "S", rel=2e-5)


def test_linearity_and_batch_independence_at_bench_width():
    """Size-independent properties at the benchmark's channel width: (a) samples of a batch are independent
    (quirk Q8): sample 0 of a batch of 2 equals the batch-of-1 result bit for bit; (b) the upsampler is affine:
    out(a) + out(b) - out(0) == out(a + b) within fp32 rounding."""
    from src.model.nets import RefineNet
    dev = _dev()
    cfg = orc.exp1_x4_config(num_updated_frames=2)
    sd = orc.init_state_dict(cfg, seed=11)
    net = RefineNet(**cfg)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    inputs, _, pos = orc.synthetic_batch(cfg, n=2, t=1, h=32, w=32, seed=12)
    with torch.no_grad():
        both = net([x.to(dev) for x in inputs], pos.to(dev))[-1][0]
        one = net([x[:1].to(dev) for x in inputs], pos[:1].to(dev))[-1][0]
    assert torch.equal(both[:1], one)


def test_src_main_trains_from_yaml(tmp_path):
    """python -m src.main <yaml>: one epoch of training + validation on the synthetic cines, checkpoint written."""
    import types
    import yaml
    from conftest import PKG
    from src import main as M
    cfg = yaml.safe_load(open(os.path.join(PKG, 'configs', 'refine_net_x4_synthetic.yaml')))
    cfg['main']['saved_dir'] = str(tmp_path / 'run')
    cfg['trainer']['kwargs']['num_epochs'] = 1
    cfg['dataloader']['kwargs']['num_workers'] = 0
    cfg['monitor']['kwargs']['saved_freq'] = 1
    p = tmp_path / 'cfg.yaml'
    p.write_text(yaml.safe_dump(cfg))
    M.main(types.SimpleNamespace(config_path=p, test=False))
    ck = torch.load(tmp_path / 'run' / 'checkpoints' / 'model_1.pth', map_location='cpu', weights_only=False)
    assert set(ck) == {'net', 'optimizer', 'lr_scheduler', 'monitor', 'epoch', 'random_state', 'np_random_seeds'}
    assert list(ck['net'].keys()) == list(orc.state_dict_spec(orc.exp1_x4_config()).keys())
    log = (tmp_path / 'run' / 'log' / 'scalars.jsonl').read_text()
    assert '"Loss"' in log and '"PSNR"' in log


@pytest.mark.parametrize('name,over,n,t,h,w', [
    ('cfg4-like x2 T=5', dict(upscale_factor=2), 2, 5, 40, 24),
    ('cfg5-like x4 phase code T=11', dict(), 1, 11, 48, 48),
    ('x3', dict(upscale_factor=3, num_stages=2), 1, 2, 33, 20),
])
def test_other_baseline_configs_vs_oracle(name, over, n, t, h, w):
    """BASELINE.json configs 4 and 5 (and x3) at full channel width and reduced spatial size, against the CPU oracle on
    the same seeded inputs: all outputs, the loss and every parameter gradient."""
    cfg = orc.exp1_x4_config(**over)
    sd = orc.init_state_dict(cfg, seed=41)
    inputs, targets, pos = orc.synthetic_batch(cfg, n, t, h, w, seed=42)
    torch.set_num_threads(16)
    ref_out, ref_loss, ref_grads = orc.step(sd, cfg, [x.clone() for x in inputs], targets, pos)
    net, _, outs, loss = _module_step(dict(cfg), sd, inputs, targets, pos, torch.nn.L1Loss())
    for go, gr in zip(outs, ref_out):
        for a, b in zip(go, gr):
            torch.testing.assert_close(a.detach().cpu(), b, atol=1e-4, rtol=1e-4)
    assert abs(float(loss.detach()) - float(ref_loss)) <= 1e-5 * abs(float(ref_loss))
    for k, p in net.named_parameters():
        if ref_grads[k] is None:
            assert p.grad is None
        else:
            _grad_close(p.grad, ref_grads[k], k)


def test_rccl_allreduce_of_flat_gradient_single_rank(g1):
    """The collective of the data-parallel path on the real backend (nccl = RCCL), world size 1: the flat gradient
    buffer the engine writes is reduced in place and the parameter gradients (views of it) are unchanged."""
    import torch.distributed as dist
    from hipvsr import dp
    c = g1['x2_pos1_mem1']
    net, _, _, _ = _module_step(c['kwargs'], c['state_dict'], c['inputs'], c['targets'], c['pos_codes'], torch.nn.L1Loss())
    before = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=_dev())
    try:
        nbytes = dp.allreduce_gradients(net, force=True)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    assert nbytes == sum(p.numel() for p in net.parameters()) * 4       # in place over the engine's flat buffer
    for k, p in net.named_parameters():
        if p.grad is not None:
            assert torch.equal(p.grad, before[k])


def test_phase_plane_sources_many_images_vs_torch():
    """Five 4-channel phase-plane sources over 30 images of 128x128 (the geometry of refine conv1 at BASELINE
    config 2 with N = 2): half of the operand loads of such a call have all 64 lanes out of range.  The hardware
    returns those ahead of older loads, which broke the counted waits of the implicit-GEMM kernel at this size
    (whole workgroups summed stale registers; small shapes never showed it)."""
    import torch.nn.functional as F
    from hipvsr.hip_ops import HipOps
    from hipvsr.plans import Dst, NetPlans, Src
    from hipvsr.spec import NetConfig, state_dict_spec
    dev = _dev()
    cfg = NetConfig(1, 1, [64, 64, 64], num_stages=3, refine_window_size=5, upscale_factor=4, update_memory=True,
                    num_updated_frames=6, positional_encoding=True)
    P, ops = NetPlans(cfg), HipOps(dev)
    assert P.r1_wino
    g = torch.Generator('cpu').manual_seed(5)
    w1 = (torch.randn(state_dict_spec(cfg)[P.r1_fwd_p.wkey], generator=g) * 0.05).to(dev)
    ops.pack(P.r1_fwd_p, w1, None)
    N, H, W, Fr = 2, 128, 128, 19
    B = (Fr - 4) * N
    P4 = torch.zeros(Fr * N, H, W, 4, device=dev)
    P4[..., 0] = torch.randn(Fr * N, 1, 1, generator=g).to(dev)
    out = torch.zeros(B, H, W, 132, device=dev)
    ops.conv(P.r1_fwd_p, [Src(P4, img_off=j * N) for j in range(5)], B, H, W, dsts=[Dst(out, 128, accumulate=True)])
    torch.cuda.synchronize()
    x = torch.cat([P4[j * N:j * N + B, ..., :1] for j in range(5)], -1).permute(0, 3, 1, 2).cpu()
    ref = F.conv2d(x, w1[:128, [j * 129 + 128 for j in range(5)]].cpu(), None, padding=1).permute(0, 2, 3, 1)
    torch.testing.assert_close(out[..., :128].cpu(), ref, atol=1e-5, rtol=1e-5)


def test_training_step_is_bitwise_repeatable():
    """No atomics anywhere: the same step must give the same bits.  30 repetitions of BASELINE config 1 caught both
    races that the asm-scheduled kernels had (tests/stress/race_check.py, tools/race_kernel.py are the long versions)."""
    from src.model.nets import RefineNet
    dev = _dev()
    cfg = orc.exp1_x4_config()
    net = RefineNet(**cfg)
    net.load_state_dict(orc.init_state_dict(cfg, seed=1))
    net = net.to(dev).train()
    inputs, targets, pos = orc.synthetic_batch(cfg, 1, 3, 64, 64, seed=2)
    xs, ys, pc = [x.to(dev) for x in inputs], [y.to(dev) for y in targets], pos.to(dev)
    ref = None
    for r in range(30):
        net.zero_grad()
        outs = net(xs, pc)
        sum((o - y).abs().mean() for grp in outs for o, y in zip(grp, ys)).backward()
        torch.cuda.synchronize()
        cur = [o.detach().clone() for grp in outs for o in grp] + [p.grad.clone() for p in net.parameters() if p.grad is not None]
        if ref is None:
            ref = cur
        else:
            assert all(torch.equal(a, b) for a, b in zip(cur, ref)), r


def test_last_group_only_inference(g1):
    """`net.last_group_only = True` under no_grad: outputs[-1] is bit-identical to the full forward, the other groups are
    not computed (None); with gradients enabled the flag is ignored."""
    from src.model.nets import RefineNet
    dev = _dev()
    c = g1['x4_pos1_mem1']
    net = RefineNet(**c['kwargs'])
    net.load_state_dict(c['state_dict'])
    net = net.to(dev).eval()
    xs, pc = [x.to(dev) for x in c['inputs']], c['pos_codes'].to(dev)
    with torch.no_grad():
        full = net(xs, pc)
        net.last_group_only = True
        last = net(xs, pc)
    assert all(g is None for g in last[:-1]) and len(last) == len(full)
    for a, b in zip(last[-1], full[-1]):
        assert torch.equal(a, b)
    net.train()
    out = net(xs, pc)
    assert all(g is not None for g in out)


@pytest.mark.parametrize('W', [32, 16, 64])
@pytest.mark.parametrize('which', ['lstm', 'up', 'refine'])
def test_winograd_weight_gradient_vs_pixel_contraction(which, W):
    """rnh_wino_wgrad (F(3x3,2x2): padded gather, transforms per lane, G^T.G in the reduction) against rnh_conv_wgrad
    on the same operands: ConvLSTM (two 64-channel sources, 256 columns, bias), PixelShuffle conv (dy gathered from the
    2x larger tensor, strided column map) and refine conv1's hidden-state rows (ten sources with frame offsets).
    W = 32, 64: the variant that shares the input transform through LDS (quads of groups); W = 16: the per-lane kernel.
    Both accumulate the tiles in the same order: where both apply they must agree bit for bit."""
    from hipvsr.hip_ops import HipOps
    from hipvsr.plans import NetPlans, Src
    from hipvsr.spec import NetConfig
    dev = _dev()
    cfg = NetConfig(1, 1, [64, 64, 64], num_stages=3, refine_window_size=5, upscale_factor=4, update_memory=True,
                    num_updated_frames=6, positional_encoding=True)
    P = NetPlans(cfg)
    g = torch.Generator('cpu').manual_seed(17)
    R = lambda *s: torch.randn(*s, generator=g).to(dev)                       # noqa: E731
    B, H = 3, 6
    if which == 'lstm':
        plan = P.lstm[('forward', 1)]['wgrad']
        xs = [Src(R(B + 1, H, W, 64), img_off=1), Src(R(B + 1, H, W, 64))]
        ys = [Src(R(B, H, W, 256))]
        shape, bias = (256, 128, 3, 3), True
    elif which == 'up':
        plan = P.up[0]['wgrad']
        xs = [Src(R(B, H, W, 64))]
        big = R(B, 2 * H, 2 * W, 64)
        ys = [Src(big, scale=2, sub=(ij // 2, ij % 2)) for ij in range(4)]
        shape, bias = (256, 64, 3, 3), True
    else:
        plan = P.r1_wgrad_h
        Hf, Hb = R(B + 4, H, W, 64), R(B + 4, H, W, 64)
        xs = []
        for j in range(5):
            xs += [Src(Hf, img_off=j), Src(Hb, img_off=j)]
        ys = [Src(R(B, H, W, 132), nch=128)]
        shape, bias = (129, 645, 3, 3), True
    res = []
    old = os.environ.get('RNH_WGRAD_LDS')
    try:
        for wino, lds in ((True, '1'), (False, '1'), (True, '0')):
            os.environ['RNH_WGRAD_LDS'] = lds
            ops = HipOps(dev)
            ops.wino_wgrad = wino
            dw, db = torch.zeros(shape, device=dev), torch.zeros(shape[0], device=dev)
            ops.wgrad(plan, xs, ys, B, H, W, dw, db if bias else None)
            torch.cuda.synchronize()
            res.append((dw.cpu(), db.cpu()))
    finally:
        if old is None:
            os.environ.pop('RNH_WGRAD_LDS', None)
        else:
            os.environ['RNH_WGRAD_LDS'] = old
    _grad_close(res[0][0], res[1][0], 'dw', rel=1e-5)
    _grad_close(res[0][1], res[1][1], 'db', rel=1e-5)
    assert float(res[1][0].abs().max()) > 0
    assert torch.equal(res[0][0], res[2][0]) and torch.equal(res[0][1], res[2][1])      # LDS variant == per-lane variant
