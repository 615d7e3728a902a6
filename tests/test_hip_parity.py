"""Parity of the HIP path (through the C ABI) with the reference, on the GPU box.

* golden vectors captured from the reference itself (tests/golden/g1_tiny.pt, g2_cfg1.pt, g5_edges.pt);
* the oracle (oracle/refinenet_oracle.py) on seeded inputs at sizes it finishes in seconds;
* kernel-level comparisons against the torch semantics in tests/torch_ops.py on shapes that exercise tile
  tails, several M tiles, every epilogue and every tile shape.

Tolerances (fp32, K up to 5 805 re-associated products): outputs atol 1e-4 / rtol 1e-4;
gradients ELEMENTWISE |delta| <= 1e-5 + 1e-3 |g| and ||delta||_2 <= 1e-3 ||g||_2 per tensor; loss rtol 1e-5;
PSNR |delta| < 0.01 dB (north_star).  Kernel-level comparisons against a float64 torch evaluation of the same
operation are tighter: |delta| <= 1e-3 |ref| + a few fp32 ulps of the tensor's largest element.
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


def _grad_close(mine, ref, name, atol=1e-5, rtol=1e-3, l2=1e-3):
    """The contract's gradient criterion: every element within atol + rtol |g|, and the tensor's L2 error within l2."""
    a, b = mine.detach().cpu().double(), ref.detach().cpu().double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    d = (a - b).abs()
    over = d - (atol + rtol * b.abs())
    i = int(over.argmax())
    assert float(over.flatten()[i]) <= 0, (name, 'element', i, float(a.flatten()[i]), float(b.flatten()[i]), 'max|g|', float(b.abs().max()))
    assert float(d.norm()) <= l2 * float(b.norm()) + 1e-12, (name, 'L2', float(d.norm()), float(b.norm()))


def _max_close(mine, ref, name, rel):
    """Kernel-level bound relative to the tensor's largest element (fp32 summation noise scales with it)."""
    scale = float(ref.abs().max())
    err = float((mine.detach().cpu().double() - ref.double()).abs().max())
    assert err <= rel * scale + 1e-6, (name, err, scale)


def _kernel_close(mine, ref64, name, ulps=32, rtol=1e-3):
    """HIP kernel output against a float64 evaluation: |delta| <= rtol |ref| + ulps * 2^-24 * max|ref| elementwise."""
    a, b = mine.detach().cpu().double(), ref64.detach().cpu().double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    d = (a - b).abs()
    over = d - (rtol * b.abs() + ulps * 2.0 ** -24 * float(b.abs().max()))
    i = int(over.argmax())
    assert float(over.flatten()[i]) <= 0, (name, i, float(a.flatten()[i]), float(b.flatten()[i]), float(b.abs().max()))
    assert float(d.norm()) <= 1e-5 * float(b.norm()) + 1e-12, (name, 'L2', float(d.norm()), float(b.norm()))


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
            _grad_close(p.grad.flatten()[:16], r['grad_head'][k], k)
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
    _max_close(dy1, want, 'dY1', rel=2e-5)
    if ops.uptail_xcorr_supported(C1, r, 1):
        M, S = ops.uptail_xcorr(y1.to(dev), d_o.to(dev), r)
        torch.cuda.synchronize()
        Mr, Sr = ref.uptail_xcorr(y1, d_o, r)
        _max_close(M, Mr, "M", rel=2e-5)
        _max_close(S, Sr, "S", rel=2e-5)


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


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_src_main_trains_from_yaml(tmp_path, monkeypatch, dtype):
    """python -m src.main <yaml>: one epoch of training + validation on the synthetic cines, checkpoint written - in the
    reference's precision and (RNH_DTYPE=bf16 in the environment, the YAML schema has no such key) on the bf16-storage path;
    the checkpoint is fp32 with the reference's keys either way."""
    import types
    import yaml
    from conftest import PKG
    from src import main as M
    monkeypatch.setenv('RNH_DTYPE', dtype)
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
    assert all(v.dtype == torch.float32 for v in ck['net'].values())
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
    import socket
    with socket.socket() as sk:                                             # a free port: parallel runs on one host must not collide
        sk.bind(('127.0.0.1', 0))
        os.environ['MASTER_PORT'] = str(sk.getsockname()[1])
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


def _full_cfg():
    from hipvsr.spec import NetConfig
    return NetConfig(1, 1, [64, 64, 64], num_stages=3, refine_window_size=5, upscale_factor=4, update_memory=True,
                     num_updated_frames=6, positional_encoding=True)


def _nchw64(t):
    return t.detach().cpu().double().permute(0, 3, 1, 2)


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


# (B, H, W): W = 16 / 32 / 64, odd H and W, tile tails; the last two have tile grids of 8 x 4 / 16 x 8 (the kernel's blocked tile
# order: a workgroup = an 8 x 4 block of tiles)
WINO_SHAPES = [(3, 6, 16), (2, 6, 32), (2, 4, 64), (2, 5, 13), (1, 7, 34), (3, 8, 16), (2, 16, 32)]


@pytest.mark.parametrize('B,H,W', WINO_SHAPES)
@pytest.mark.parametrize('which', ['lstm', 'lstm_first', 'lstm_dgrad', 'up', 'up_dgrad', 'refine', 'refine_dgrad',
                                   'lstm@64', 'lstm_first@64', 'lstm_dgrad@64', 'refine@64'])
def test_winograd_conv_kernels_vs_torch_float64(which, B, H, W, monkeypatch):
    """rnh_conv_wino at full channel width (the 13 reference goldens use num_features [8, 8], which the plans route to the
    implicit GEMM) against float64 torch convolutions of the OIHW weights: ConvLSTM cell with the fused gate epilogue
    (two K sources / the zero-state K = 576 plan; gates_out, c', h'), its data gradient (two destinations), the
    PixelShuffle convolution (PS epilogue) and its data gradient (pixel-unshuffle fused into the staging loads),
    refine conv1 over the ten hidden-state sources and its gather-form data gradient (accumulating stores).  The plans give
    these convolutions the kernel's 128-column geometry (8 waves, one workgroup per CU); '@64' runs them in the 64-column one."""
    import torch.nn.functional as F
    from hipvsr.hip_ops import HipOps
    from hipvsr.plans import Dst, NetPlans, Src
    from hipvsr.spec import state_dict_spec
    dev = _dev()
    cfg = _full_cfg()
    if which.endswith('@64'):                      # the 64-column geometry (two workgroups per CU) where the plans would pick 128
        monkeypatch.setenv('RNH_WINO_COLS', '64')
        which = which[:-3]
    P, ops = NetPlans(cfg), HipOps(dev)
    spec = state_dict_spec(cfg)
    g = torch.Generator('cpu').manual_seed(1000 + 7 * W + H)
    R = lambda *sh: torch.randn(*sh, generator=g)                           # noqa: E731
    if which in ('lstm', 'lstm_first', 'lstm_dgrad'):
        pl = P.lstm[('backward', 2)]
        w, b = R(*spec[pl['full'].wkey]) * 0.03, R(*spec[pl['full'].bkey]) * 0.1
        wd, bd = w.to(dev), b.to(dev)
        if which == 'lstm_dgrad':
            plan = pl['dgrad']
            assert plan.wino
            ops.pack(plan, wd, None)
            dg = R(B, H, W, 256)
            dx, dh = torch.full((B, H, W, 64), float('nan'), device=dev), torch.full((B, H, W, 64), float('nan'), device=dev)
            ops.conv(plan, [Src(dg.to(dev))], B, H, W, dsts=[Dst(dx, 64), Dst(dh, 64)])
            torch.cuda.synchronize()
            ref = _nhwc(F.conv_transpose2d(_nchw64(dg), w.double(), padding=1))
            _kernel_close(dx, ref[..., :64], 'dx')
            _kernel_close(dh, ref[..., 64:], 'dh')
            return
        first = which == 'lstm_first'
        plan = pl['first'] if first else pl['full']
        assert plan.wino
        ops.pack(plan, wd, bd)
        x, h, c = R(B + 1, H, W, 64), R(B + 2, H, W, 64), R(B, H, W, 64)
        ho, co = (torch.full((B, H, W, 64), float('nan'), device=dev) for _ in range(2))
        go = torch.full((B, H, W, 256), float('nan'), device=dev)
        srcs = [Src(x.to(dev), img_off=1)] + ([] if first else [Src(h.to(dev), img_off=2)])
        ops.conv(plan, srcs, B, H, W, lstm=dict(hd=64, c_prev=None if first else c.to(dev), h_out=ho, c_out=co, gates_out=go))
        torch.cuda.synchronize()
        xin = _nchw64(x[1:]) if first else torch.cat([_nchw64(x[1:]), _nchw64(h[2:])], 1)
        pre = F.conv2d(xin, (w[:, :64] if first else w).double(), b.double(), padding=1)
        gi, gf, gop, gg = pre.split(64, dim=1)                                # reference order i, f, o, g (refine_net.py:258)
        gi, gf, gop, gg = torch.sigmoid(gi), torch.sigmoid(gf), torch.sigmoid(gop), torch.tanh(gg)
        cn = gi * gg if first else gf * _nchw64(c) + gi * gg
        _kernel_close(go, _nhwc(torch.cat([gi, gf, gop, gg], 1)), 'gates')
        _kernel_close(co, _nhwc(cn), 'c')
        _kernel_close(ho, _nhwc(gop * torch.tanh(cn)), 'h')
    elif which in ('up', 'up_dgrad'):
        u = P.up[0]
        w, b = R(256, 64, 3, 3) * 0.04, R(256) * 0.1
        if which == 'up':
            assert u['fwd'].wino
            ops.pack(u['fwd'], w.to(dev), b.to(dev))
            x = R(B, H, W, 64)
            Y = torch.full((B, 2 * H, 2 * W, 64), float('nan'), device=dev)
            ops.conv(u['fwd'], [Src(x.to(dev))], B, H, W, ps=(Y, 2))
            torch.cuda.synchronize()
            _kernel_close(Y, _nhwc(F.pixel_shuffle(F.conv2d(_nchw64(x), w.double(), b.double(), padding=1), 2)), 'Y')
        else:
            assert u['dgrad'].wino
            ops.pack(u['dgrad'], w.to(dev), None)
            dY = R(B, 2 * H, 2 * W, 64)
            dYd = dY.to(dev)
            dx = torch.full((B, H, W, 64), float('nan'), device=dev)
            ops.conv(u['dgrad'], [Src(dYd, scale=2, sub=(ij // 2, ij % 2)) for ij in range(4)], B, H, W, dsts=[Dst(dx, 64)])
            torch.cuda.synchronize()
            _kernel_close(dx, _nhwc(F.conv_transpose2d(F.pixel_unshuffle(_nchw64(dY), 2), w.double(), padding=1)), 'dx')
    else:
        assert P.r1_wino
        w1, b1 = R(129, 645, 3, 3) * 0.02, R(129) * 0.1
        hidx = [j * 129 + c for j in range(5) for c in range(128)]          # the hidden-state input channels of the 5 frame slots
        if which == 'refine':
            ops.pack(P.r1_fwd_h, w1.to(dev), b1.to(dev))
            Hf, Hb = R(B + 4, H, W, 64), R(B + 4, H, W, 64)
            Hfd, Hbd = Hf.to(dev), Hb.to(dev)
            srcs = []
            for j in range(5):
                srcs += [Src(Hfd, img_off=j), Src(Hbd, img_off=j)]
            R1 = torch.full((B, H, W, 132), float('nan'), device=dev)
            ops.conv(P.r1_fwd_h, srcs, B, H, W, dsts=[Dst(R1, 128)])
            torch.cuda.synchronize()
            xin = torch.cat([torch.cat([_nchw64(Hf[j:j + B]), _nchw64(Hb[j:j + B])], 1) for j in range(5)], 1)
            _kernel_close(R1[..., :128], _nhwc(F.conv2d(xin, w1[:128, hidx].double(), b1[:128].double(), padding=1)), 'R1')
            assert bool(torch.isnan(R1[..., 128:]).all())                    # columns beyond the destination stay untouched
        else:
            ops.pack(P.r1_dgrad_h, w1.to(dev), None)
            gs = R(B + 4, H, W, 132)
            base_f, base_b = R(B, H, W, 64), R(B, H, W, 64)
            dHf, dHb = base_f.to(dev), base_b.to(dev)
            gsd = gs.to(dev)
            ops.conv(P.r1_dgrad_h, [Src(gsd, nch=128, img_off=4 - j) for j in range(5)], B, H, W,
                     dsts=[Dst(dHf, 64, accumulate=True), Dst(dHb, 64, accumulate=True)])
            torch.cuda.synchronize()
            tot = 0
            for j in range(5):                                               # frame f collects from window f + 2 - j, slot j
                tot = tot + F.conv_transpose2d(_nchw64(gs[4 - j:4 - j + B, ..., :128]), w1[:128, j * 129:j * 129 + 128].double(), padding=1)
            tot = _nhwc(tot)
            _kernel_close(dHf, base_f.double() + tot[..., :64], 'dHf')
            _kernel_close(dHb, base_b.double() + tot[..., 64:], 'dHb')


@pytest.mark.parametrize('B,H,W', [(3, 6, 32), (3, 6, 16), (2, 4, 64), (2, 5, 13), (1, 6, 96)])
@pytest.mark.parametrize('which', ['lstm', 'up', 'refine'])
def test_weight_gradient_kernels_vs_torch_float64(which, B, H, W):
    """Weight / bias gradients at full channel width against float64 autograd of conv2d: rnh_wino_wgrad (F(3x3, 2x2);
    W % 32 == 0: the variant that shares the input transform through LDS, W = 16: the per-lane variant on the padded
    copy) and, where the Winograd form does not apply (odd sizes), rnh_conv_wgrad.  ConvLSTM (two 64-channel sources,
    256 columns), PixelShuffle conv (dy gathered from the 2x larger tensor, strided column map) and refine conv1's
    hidden-state rows (ten sources with frame offsets).  Where the Winograd variants apply (per lane, LDS-shared with one workgroup
    per CU, LDS-shared with half the transform domain per workgroup) they accumulate the tiles in the same order and must agree bit for bit."""
    import torch.nn.functional as F
    from hipvsr.hip_ops import HipOps
    from hipvsr.plans import NetPlans, Src
    dev = _dev()
    P = NetPlans(_full_cfg())
    g = torch.Generator('cpu').manual_seed(17 + W)
    R = lambda *sh: torch.randn(*sh, generator=g)                           # noqa: E731

    def ref_wgrad(x_nchw, dy_nchw, cout, cin):
        w0 = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv2d(x_nchw, w0, padding=1).backward(dy_nchw)
        return w0.grad, dy_nchw.sum(dim=(0, 2, 3))

    if which == 'lstm':
        plan = P.lstm[('forward', 1)]['wgrad']
        x, h, dy = R(B + 1, H, W, 64), R(B + 1, H, W, 64), R(B, H, W, 256)
        xs = [Src(x.to(dev), img_off=1), Src(h.to(dev))]
        ys = [Src(dy.to(dev))]
        shape = (256, 128, 3, 3)
        rw, rb = ref_wgrad(torch.cat([_nchw64(x[1:]), _nchw64(h[:B])], 1), _nchw64(dy), 256, 128)
        sel = None
    elif which == 'up':
        plan = P.up[0]['wgrad']
        x, big = R(B, H, W, 64), R(B, 2 * H, 2 * W, 64)
        xs = [Src(x.to(dev))]
        bigd = big.to(dev)
        ys = [Src(bigd, scale=2, sub=(ij // 2, ij % 2)) for ij in range(4)]
        shape = (256, 64, 3, 3)
        rw, rb = ref_wgrad(_nchw64(x), F.pixel_unshuffle(_nchw64(big), 2), 256, 64)
        sel = None
    else:
        plan = P.r1_wgrad_h
        Hf, Hb, dy = R(B + 4, H, W, 64), R(B + 4, H, W, 64), R(B, H, W, 132)
        Hfd, Hbd = Hf.to(dev), Hb.to(dev)
        xs = []
        for j in range(5):
            xs += [Src(Hfd, img_off=j), Src(Hbd, img_off=j)]
        ys = [Src(dy.to(dev), nch=128)]
        shape = (129, 645, 3, 3)
        xin = torch.cat([torch.cat([_nchw64(Hf[j:j + B]), _nchw64(Hb[j:j + B])], 1) for j in range(5)], 1)
        g640, rb128 = ref_wgrad(xin, _nchw64(dy[..., :128]), 128, 640)
        hidx = [j * 129 + c for j in range(5) for c in range(128)]
        rw = torch.zeros(shape, dtype=torch.float64)
        rw[:128, hidx] = g640
        rb = torch.zeros(129, dtype=torch.float64)
        rb[:128] = rb128
        sel = True
    res = {}
    old = {k: os.environ.get(k) for k in ('RNH_WGRAD_LDS', 'RNH_WGRAD_HALF', 'RNH_WINO44F_WGRAD')}
    try:
        os.environ['RNH_WINO44F_WGRAD'] = '0'                 # (the F(2x2)-tile variants: the fused F(4x4)-tile form has its own test, tests/test_wino44.py)
        # 'half' = the product's choice where W % 32 == 0 (two workgroups per CU, each half of the transform domain), 'lds' = one workgroup
        # per CU with all 16 positions, 'lane' = no LDS sharing, 'pixel' = the direct (non-Winograd) kernel
        for name, wino, lds, half in (('half', True, '1', '1'), ('lds', True, '1', '0'), ('lane', True, '0', '1'), ('pixel', False, '1', '1')):
            os.environ['RNH_WGRAD_LDS'], os.environ['RNH_WGRAD_HALF'] = lds, half
            ops = HipOps(dev)
            ops.wino_wgrad = wino
            dw, db = torch.zeros(shape, device=dev), torch.zeros(shape[0], device=dev)
            ops.wgrad(plan, xs, ys, B, H, W, dw, db)
            torch.cuda.synchronize()
            res[name] = (dw.cpu(), db.cpu())
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    for name, (dw, db) in res.items():
        _kernel_close(dw, rw, f'{which}.dw[{name}]', ulps=64)
        _kernel_close(db, rb, f'{which}.db[{name}]', ulps=64)
        if sel:
            mask = torch.ones(shape, dtype=torch.bool)
            mask[:128, hidx] = False
            assert float(dw[mask].abs().max()) == 0.0                      # entries the plan does not map stay untouched
    assert torch.equal(res['lds'][0], res['lane'][0]) and torch.equal(res['lds'][1], res['lane'][1])
    assert torch.equal(res['half'][0], res['lds'][0]) and torch.equal(res['half'][1], res['lds'][1])


def _cfg2_step(n, seed=202):
    cfg = orc.exp1_x4_config()
    sd = orc.init_state_dict(cfg, seed=seed)
    inputs, targets, pos = orc.synthetic_batch(cfg, n, 7, 128, 128, seed=seed + 1)
    return cfg, sd, inputs, targets, pos


@pytest.fixture(params=['0', 'force', 'force+refine2'], ids=['cell_f2x2', 'cell_f4x4', 'cell_f4x4_refine2_f4x4'])
def cell_form(request, monkeypatch):
    """The two forms of the fp32 ConvLSTM cell forward: Winograd F(2x2, 3x3) (rnh_conv_wino; RNH_WINO44=0) and F(4x4, 3x3) (rnh_wino44_cell; the engine's
    choice wherever the images are whole 4x4 tiles); the third case adds the opt-in F(4x4) form of refine conv2's forward / data gradient, whose
    transformed input also feeds conv2's weight gradient (RNH_WINO44_REFINE2=1)."""
    monkeypatch.setenv('RNH_WINO44', request.param.split('+')[0])
    if '+' in request.param:
        monkeypatch.setenv('RNH_WINO44_REFINE2', '1')
    return request.param


def test_cfg2_geometry_vs_oracle(cell_form):
    """BASELINE config 2's geometry - exp1_x4 net, T = 7 (F = 19), 128x128 -> 512x512 - at N = 2 against the CPU oracle
    (== the reference, reference acdc_vsr_refinenet_trainer.py:42-46, :83-94): all 63 outputs, the training loss, every
    parameter gradient under the contract's elementwise + L2 criterion, and PSNR |delta| < 0.01 dB.  (Refine conv1 was
    once wrong by 1.5e-3 at exactly this size and at no smaller one: DESIGN.md section 4.)"""
    import functools
    from oracle import step_tail_oracle as sto
    from src.model.metrics import PSNR
    from src.utils import denormalize
    cfg, sd, inputs, targets, pos = _cfg2_step(2)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref_out, ref_loss, ref_grads = orc.step(sd, cfg, [x.clone() for x in inputs], targets, pos)
    net, tr, outs, loss = _module_step(dict(cfg), sd, inputs, targets, pos, torch.nn.L1Loss())
    assert len(outs) == 9 and all(len(grp) == 7 for grp in outs)
    worst = 0.0
    for go, gr in zip(outs, ref_out):
        for a, b in zip(go, gr):
            assert tuple(a.shape) == (2, 1, 512, 512)
            torch.testing.assert_close(a.detach().cpu(), b, atol=1e-4, rtol=1e-4)
            worst = max(worst, float((a.detach().cpu() - b).abs().max()))
    assert abs(float(loss.detach()) - float(ref_loss)) <= 1e-5 * abs(float(ref_loss)), (float(loss), float(ref_loss))
    for k, p in net.named_parameters():
        if ref_grads[k] is None:
            assert p.grad is None
        else:
            _grad_close(p.grad, ref_grads[k], k)
    tr.metric_fns = [PSNR().to(_dev())]
    tr._denormalize = functools.partial(denormalize, dataset='acdc')
    psnr = float(tr._compute_metrics(outs, [t.to(_dev()) for t in targets])[0])
    want = float(sto.trainer_metrics(ref_out[-1], targets)[0])
    assert abs(psnr - want) < 0.01, (psnr, want)
    print(f'cfg2 geometry N=2: max |output - oracle| = {worst:.3e}, loss {float(loss):.7f} vs {float(ref_loss):.7f}, PSNR {psnr:.4f} vs {want:.4f}')


def test_bench_launch_geometry_n8_equals_replicated_n2(cell_form):
    """The exact launch geometry of the benchmark (N = 8, T = 7, 128x128): a batch made of four copies of the N = 2
    batch of the test above.  Samples are independent bit for bit (quirk Q8), so every one of the 8 samples of every
    output of the forward must equal its N = 2 twin exactly - which pins all 2048-workgroup cell launches, the 45-window
    refine launch and the 168-image upsampler launches to values that are checked against the oracle; the loss is a mean
    over samples, so the gradients of the replicated batch equal the N = 2 gradients (summation order differs: contract
    tolerance)."""
    cfg, sd, inputs, targets, pos = _cfg2_step(2)
    net2, _, outs2, loss2 = _module_step(dict(cfg), sd, inputs, targets, pos, torch.nn.L1Loss())
    g2 = {k: p.grad.detach().cpu().clone() for k, p in net2.named_parameters() if p.grad is not None}
    o2 = [[o.detach().clone() for o in grp] for grp in outs2]
    del net2, outs2
    rep = lambda t: torch.cat([t] * 4, 0)                                   # noqa: E731
    net8, _, outs8, loss8 = _module_step(dict(cfg), sd, [rep(x) for x in inputs], [rep(t) for t in targets], rep(pos), torch.nn.L1Loss())
    for ga, gb in zip(outs8, o2):
        for a, b in zip(ga, gb):
            assert tuple(a.shape) == (8, 1, 512, 512)
            for q in range(4):
                assert torch.equal(a[2 * q:2 * q + 2], b), q
    assert abs(float(loss8) - float(loss2)) <= 1e-6 * abs(float(loss2))
    for k, p in net8.named_parameters():
        if k in g2:
            _grad_close(p.grad, g2[k], k)


def test_bench_timed_code_path_first_step_loss_vs_oracle(monkeypatch):
    """the code path bench.py TIMES - bench.run_case at BASELINE config 2 (its net, its synthetic batch, its trainer step, the
    forms the engine resolves by itself), one step, no warm-up - is itself pinned to the oracle: the step's loss (`config.final_loss`, the training
    loss of the N = 8 batch at the initial weights) against the CPU oracle's (== reference, acdc_vsr_refinenet_trainer.py:83-94) on the same net and
    batch, evaluated two samples at a time (every term is a mean over samples: the mean of the four pair losses), 1e-5 relative; and the line carries
    the resolved forms.  (A second step would need the oracle's backward at N = 8 on the CPU: minutes; the trajectory tests of test_parity_r04.py
    cover the optimizer's side at a smaller shape.)"""
    import bench
    from hipvsr import forms
    for k in forms.SWITCHES:
        monkeypatch.delenv(k, raising=False)
    dev = _dev()
    args = bench.parse_args(['--steps', '1', '--warmup', '0', '--no-secondary', '--no-cpu-baseline'])
    out = bench.run_case(args, 'f32', dev, 1, 0)
    fm = out['config']['forms']
    assert 'F(4x4,3x3)' in fm['cell'] and fm['env_overrides'] == [] and fm['gates'] == 'stored' and out['roofline']['frac_with_transform'] < out['roofline']['frac']
    net = bench.make_net(dev, seed=0, scale=4)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    inputs, targets, pos = bench.synthetic_batch(dev, args.batch, args.frames, args.size, args.size, seed=20200526 + 2, s=4)
    inputs, targets, pos = [x.cpu() for x in inputs], [y.cpu() for y in targets], pos.cpu()
    del net
    cfg = orc.exp1_x4_config()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    losses = []
    with torch.no_grad():
        for q in range(0, args.batch, 2):
            outs = orc.forward(orc.as_leaf_params(sd), cfg, [x[q:q + 2].clone() for x in inputs], pos[q:q + 2])
            losses.append(float(orc.training_loss(outs, [y[q:q + 2] for y in targets])))
    want = sum(losses) / len(losses)
    got = out['config']['final_loss']
    assert abs(got - want) <= 1e-5 * abs(want) + 1e-6, (got, want)          # (+ the 6 decimals the line rounds to)
    print(f'bench.run_case, config 2, first step: loss {got:.6f}, oracle {want:.6f}')
