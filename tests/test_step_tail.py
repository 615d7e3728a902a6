"""Rows f3 / f4 of SURVEY.md section 8: the Adam update on flat buffers and the fused PSNR / SSIM metrics.

CPU: the oracle (oracle/step_tail_oracle.py) against the golden vectors produced by the reference's own metrics.py /
utils.py / trainer and by torch.optim.Adam (tests/golden/g7_metrics.pt), host-side logic, error behaviour.
GPU (-m gpu): the HIP kernels through the C ABI against the goldens and the oracle.

Tolerances (fp32): PSNR |delta| <= 1e-4 dB (the squared errors of denormalised images are integers: sums are exact, the
difference is log10 rounding); SSIM |delta| <= 2e-5 (the windowed second moments cancel: E[x^2] - mu^2 with values up to
255^2, re-associated by the separable window); Adam: parameters after 6 steps rtol 1e-6 / atol 1e-8 (the update is
~1e-4 of the parameter, 1 ulp of sqrt / division order)."""
import math
import os

import pytest
import torch

from conftest import GOLDEN
from oracle import step_tail_oracle as sto

CASES = ['random', 'smooth', 'wide', 'minimal', 'tall']


@pytest.fixture(scope='module')
def g7():
    return torch.load(os.path.join(GOLDEN, 'g7_metrics.pt'))


# ------------------------------------------------------------------------------------------------------------------
# CPU: oracle pinned by the reference's outputs
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('case', CASES)
def test_oracle_metrics_match_the_reference_bit_for_bit(g7, case):
    c = g7[case]
    m = sto.trainer_metrics(c['outputs_last'], c['targets'])
    assert torch.equal(m[0], c['trainer_psnr']) and torch.equal(m[1], c['trainer_ssim'])
    den_o = [sto.denormalize(o) for o in c['outputs_last']]
    den_t = [sto.denormalize(t) for t in c['targets']]
    assert torch.equal(torch.stack([sto.psnr(a, b, size_average=False) for a, b in zip(den_o, den_t)]), c['psnr_per_sample'])
    assert torch.equal(torch.stack([sto.ssim(a, b, size_average=False) for a, b in zip(den_o, den_t)]), c['ssim_per_sample'])
    pm = sto.predictor_metrics(c['outputs_last'], c['targets'])
    assert pm.shape == (len(c['targets']), 2)
    assert torch.allclose(pm.mean(0), torch.stack(m), atol=1e-6)


def test_oracle_window_and_unit_range(g7):
    assert torch.equal(sto.ssim_window(2, 1), g7['ssim_weight'])
    u = g7['unit_range']
    assert torch.equal(sto.psnr(u['output'], u['target'], max_value=1), u['psnr'])
    assert torch.equal(sto.ssim(u['output'], u['target'], value_range=1), u['ssim'])
    # the separable window the kernel receives reproduces the reference's 2-D window
    from hipvsr.step_tail import ssim_window_1d
    w = torch.tensor(list(ssim_window_1d()), dtype=torch.float64)
    assert abs(float(w.sum()) - 1) < 1e-7
    assert torch.allclose(torch.outer(w, w).float(), g7['ssim_weight'][0, 0], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize('name', ['yaml', 'decay'])
def test_oracle_adam_matches_torch_optim_adam(g7, name):
    a = g7['adam'][name]
    kw = dict(a['kwargs'])
    ps = [p.clone() for p in a['p0']]
    ms, vs = [torch.zeros_like(p) for p in ps], [torch.zeros_like(p) for p in ps]
    live = [torch.nn.Parameter(p.clone()) for p in a['p0']]
    opt = torch.optim.Adam(live, **kw)                              # the class the reference instantiates (src/main.py:76)
    for k, (gs, want) in enumerate(zip(a['grads'], a['trajectory'])):
        gs = [g if i != 2 else None for i, g in enumerate(gs)]      # parameter 2 never receives a gradient
        sto.adam_step(ps, gs, ms, vs, k + 1, **kw)
        for p_, g_ in zip(live, gs):
            p_.grad = None if g_ is None else g_.clone()
        opt.step()
        for p, w, l in zip(ps, want, live):
            assert torch.equal(l.detach(), w)                       # torch.optim.Adam reproduces its recorded trajectory
            assert torch.allclose(p, w, rtol=1e-6, atol=1e-9), (k, float((p - w).abs().max()))
    assert torch.equal(ps[2], a['p0'][2])


def test_host_side_errors_without_a_gpu():
    from hipvsr import lib as L
    from hipvsr.step_tail import FlatAdam, psnr_ssim
    from src.model.metrics import PSNR, SSIM, fused_metrics
    p = torch.nn.Parameter(torch.zeros(4))
    for bad in (dict(lr=-1.0), dict(eps=-1.0), dict(betas=(1.0, 0.9)), dict(betas=(0.9, 1.0)), dict(weight_decay=-0.1), dict(amsgrad=True)):
        with pytest.raises(ValueError):
            FlatAdam([p], **bad)
    opt = FlatAdam([p], lr=1e-3)
    assert set(opt.param_groups[0]) >= set(torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))]).param_groups[0]) - {'params'}
    p.grad = torch.ones(4)
    with pytest.raises(L.HipKernelError, match='no CPU path'):
        opt.step()
    x = torch.zeros(1, 1, 16, 16)
    with pytest.raises(L.HipKernelError, match='no CPU path'):
        PSNR()(x, x)
    with pytest.raises(L.HipKernelError, match='no CPU path'):
        SSIM()(x, x)
    with pytest.raises(L.HipKernelError):
        psnr_ssim(x, x, 1, 1, 16, 16)
    with pytest.raises(ValueError, match='dim=4'):
        SSIM(dim=4)
    assert 'weight' in SSIM().state_dict() and SSIM().state_dict()['weight'].shape == (1, 1, 11, 11)
    assert fused_metrics([x], [x], [PSNR()]) is None                # CPU tensors: not served, the caller decides
    lib = L.load()
    assert lib.rnh_metrics_ws_floats(56, 512, 512) == 2 * 56 * 32 * 8
    assert lib.rnh_adam_step(None, None, None, None, 4, 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, None) != 0
    assert b'null' in lib.rnh_last_error()


def test_main_routes_adam_to_the_flat_optimizer_only_on_hip():
    from src.main import Cfg, _get_optimizer
    p = [torch.nn.Parameter(torch.zeros(3))]
    cfg = Cfg(dict(name='Adam', kwargs=dict(lr=1e-4, weight_decay=0)))
    assert type(_get_optimizer(cfg, p, torch.device('cpu'))) is torch.optim.Adam
    from hipvsr.step_tail import FlatAdam
    opt = _get_optimizer(cfg, p, torch.device('cuda:0'))            # construction does not touch the device
    assert type(opt) is FlatAdam and opt.defaults['lr'] == 1e-4
    assert type(_get_optimizer(Cfg(dict(name='Adam', kwargs=dict(amsgrad=True))), p, torch.device('cuda:0'))) is torch.optim.Adam
    assert type(_get_optimizer(Cfg(dict(name='SGD', kwargs=dict(lr=0.1))), p, torch.device('cuda:0'))) is torch.optim.SGD


# ------------------------------------------------------------------------------------------------------------------
# GPU: the kernels through the C ABI
# ------------------------------------------------------------------------------------------------------------------
def _dev():
    return torch.device('cuda:0')


@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES)
def test_fused_metrics_vs_reference_golden(g7, case):
    from src.model.metrics import PSNR, SSIM, fused_metrics
    c = g7[case]
    outs, tgts = [o.to(_dev()) for o in c['outputs_last']], [t.to(_dev()) for t in c['targets']]
    m = fused_metrics(outs, tgts, [PSNR(), SSIM()])
    assert abs(float(m[0]) - float(c['trainer_psnr'])) <= 1e-4
    assert abs(float(m[1]) - float(c['trainer_ssim'])) <= 2e-5
    m2 = fused_metrics(outs, tgts, [SSIM(), PSNR()])                 # order follows metric_fns
    assert float(m2[0]) == float(m[1]) and float(m2[1]) == float(m[0])
    only = fused_metrics(outs, tgts, [PSNR()])                       # PSNR alone skips the windowed sums
    assert float(only[0]) == float(m[0])
    pf = fused_metrics(outs, tgts, [PSNR(), SSIM()], per_frame=True).cpu()
    assert torch.allclose(pf[:, 0], c['psnr_per_sample'].mean(1), atol=1e-4, rtol=0)
    assert torch.allclose(pf[:, 1], c['ssim_per_sample'].mean(1), atol=2e-5, rtol=0)
    # the module boundary (already denormalised images), per sample
    den_o = [sto.denormalize(o).to(_dev()) for o in c['outputs_last']]
    den_t = [sto.denormalize(t).to(_dev()) for t in c['targets']]
    for i in range(len(den_o)):
        assert torch.allclose(PSNR(size_average=False)(den_o[i], den_t[i]).cpu(), c['psnr_per_sample'][i], atol=1e-4, rtol=0)
        assert torch.allclose(SSIM(size_average=False)(den_o[i], den_t[i]).cpu(), c['ssim_per_sample'][i], atol=2e-5, rtol=0)


@pytest.mark.gpu
def test_metric_modules_unit_range_channels_and_errors(g7):
    from hipvsr import lib as L
    from src.model.metrics import PSNR, SSIM
    u = g7['unit_range']
    o, t = u['output'].to(_dev()), u['target'].to(_dev())
    assert abs(float(PSNR(max_value=1)(o, t)) - float(u['psnr'])) <= 1e-4
    assert abs(float(SSIM(value_range=1)(o, t)) - float(u['ssim'])) <= 2e-5
    g = torch.Generator('cpu').manual_seed(3)
    a, b = torch.rand(3, 2, 19, 70, generator=g) * 255, torch.rand(3, 2, 19, 70, generator=g) * 255
    assert torch.allclose(SSIM(channels=2, size_average=False)(a.to(_dev()), b.to(_dev())).cpu(),
                          sto.ssim(a, b, channels=2, size_average=False), atol=2e-5, rtol=0)
    assert torch.allclose(PSNR(size_average=False)(a.to(_dev()), b.to(_dev())).cpu(), sto.psnr(a, b, size_average=False), atol=1e-4, rtol=0)
    v, w = torch.rand(4, 5, generator=g), torch.rand(4, 5, generator=g)         # (N, C) only: PSNR is defined, SSIM is not
    assert torch.allclose(PSNR(max_value=1, size_average=False)(v.to(_dev()), w.to(_dev())).cpu(), sto.psnr(v, w, 1, False), atol=1e-4, rtol=0)
    with pytest.raises(L.HipKernelError, match='smaller than the 11x11 window'):
        SSIM()(torch.zeros(1, 1, 10, 30, device=_dev()), torch.zeros(1, 1, 10, 30, device=_dev()))
    # identical images: mse = 0 -> 10 log10(255^2 / 1e-10), SSIM = 1
    assert abs(float(PSNR()(o, o)) - 10 * math.log10(255 ** 2 / 1e-10)) < 1e-3 and abs(float(SSIM()(o, o)) - 1) < 1e-6


@pytest.mark.gpu
def test_fused_metrics_at_bench_size_vs_oracle():
    """BASELINE config 2's step: 7 frames x 8 samples of 512x512 (the trainer's per-step call)."""
    from src.model.metrics import PSNR, SSIM, fused_metrics
    g = torch.Generator('cpu').manual_seed(11)
    T, N, H = 7, 8, 512
    base = torch.nn.functional.avg_pool2d(torch.randn(T * N, 1, H + 8, H + 8, generator=g), 9, stride=1) * 2.5
    tg = base.view(T, N, 1, H, H)
    out = tg + 0.03 * torch.randn(T, N, 1, H, H, generator=g)
    want = sto.trainer_metrics(list(out), list(tg))
    packed = out.to(_dev()).permute(0, 1, 3, 4, 2).reshape(T * N, H, H, 1).contiguous()
    outs = [packed[i * N:(i + 1) * N].permute(0, 3, 1, 2) for i in range(T)]
    m = fused_metrics(outs, [t.to(_dev()) for t in tg], [PSNR(), SSIM()], packed_last=packed)
    assert abs(float(m[0]) - float(want[0])) <= 1e-4 and abs(float(m[1]) - float(want[1])) <= 2e-5
    again = fused_metrics(outs, [t.to(_dev()) for t in tg], [PSNR(), SSIM()])          # adjacency found without the hint
    assert float(again[0]) == float(m[0]) and float(again[1]) == float(m[1])          # fixed-order reductions: bitwise repeatable


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['yaml', 'decay'])
def test_flat_adam_vs_torch_adam_trajectory(g7, name):
    from hipvsr.step_tail import FlatAdam
    a = g7['adam'][name]
    ps = [torch.nn.Parameter(p.clone().to(_dev())) for p in a['p0']]
    opt = FlatAdam(ps, **a['kwargs'])
    for k, (gs, want) in enumerate(zip(a['grads'], a['trajectory'])):
        for i, (p, g) in enumerate(zip(ps, gs)):
            p.grad = None if i == 2 else g.to(_dev())
        opt.step()
        for p, w in zip(ps, want):
            assert torch.allclose(p.detach().cpu(), w, rtol=1e-6, atol=1e-8), (k, float((p.detach().cpu() - w).abs().max()))
    assert torch.equal(ps[2].detach().cpu(), a['p0'][2]) and ps[2] not in opt.state
    assert int(opt.state[ps[0]]['step']) == 6


@pytest.mark.gpu
def test_flat_adam_on_the_network_two_launches_and_checkpoint_format():
    """RefineNet's own parameter list: gradients are views of the engine's flat buffer, so a step is two launches (the run
    before and the run after refine_block.prelu.weight, which never gets a gradient); the state_dict goes into a
    torch.optim.Adam and back."""
    from hipvsr.step_tail import FlatAdam
    from oracle import refinenet_oracle as orc
    from src.model.nets import RefineNet
    from src.runner.trainers import AcdcVSRRefineNetTrainer
    cfg = orc.Config(in_channels=1, out_channels=1, num_features=[8, 8], num_stages=2, refine_window_size=5, upscale_factor=2,
                     update_memory=True, num_updated_frames=2, positional_encoding=True)
    sd = orc.init_state_dict(cfg, seed=3)
    inputs, targets, pos = orc.synthetic_batch(cfg, n=2, t=2, h=8, w=8, seed=4)
    net = RefineNet(**cfg)
    net.load_state_dict(sd)
    opt = FlatAdam(net.parameters(), lr=1e-3)                       # built before .to(device), like src/main.py does
    net = net.to(_dev()).train()
    tr = object.__new__(AcdcVSRRefineNetTrainer)
    tr.net, tr.loss_fns, tr.metric_fns, tr.optimizer = net, [torch.nn.L1Loss()], [], opt
    tr.loss_weights = torch.tensor([1.0], device=_dev())
    cpu = {k: v.clone() for k, v in sd.items()}
    names = [k for k, _ in net.named_parameters()]
    ms, vs = {k: torch.zeros_like(v) for k, v in cpu.items()}, {k: torch.zeros_like(v) for k, v in cpu.items()}
    for step in range(1, 4):
        _, ref_loss, ref_grads = orc.step(cpu, cfg, [x.clone() for x in inputs], targets, pos)
        sto.adam_step([cpu[k] for k in names], [ref_grads[k] for k in names], [ms[k] for k in names], [vs[k] for k in names], step, lr=1e-3)
        _, loss, _ = tr.train_step([x.to(_dev()) for x in inputs], [t.to(_dev()) for t in targets], pos.to(_dev()))
        assert opt.launches == 2
        assert abs(float(loss) - float(ref_loss)) <= 1e-4 * abs(float(ref_loss))
        for k, p in net.named_parameters():
            # lr = 1e-3 steps on gradients that agree to 1e-3 relative: parameters agree to ~1e-6 absolute
            assert torch.allclose(p.detach().cpu(), cpu[k], rtol=0, atol=2e-5), (step, k, float((p.detach().cpu() - cpu[k]).abs().max()))
    # parameters are views of one buffer, in order
    ptrs = [p.data_ptr() for p in net.parameters()]
    assert all(b - a == 4 * p.numel() for a, b, p in zip(ptrs, ptrs[1:], net.parameters()))
    # checkpoint format: FlatAdam -> torch.optim.Adam -> FlatAdam
    sd_opt = opt.state_dict()
    ref_opt = torch.optim.Adam([torch.nn.Parameter(p.detach().cpu().clone()) for p in net.parameters()], lr=1e-3)
    ref_opt.load_state_dict(sd_opt)
    assert len(ref_opt.state) == len(names) - 1
    opt2 = FlatAdam(net.parameters(), lr=5e-4)
    opt2.load_state_dict(ref_opt.state_dict())
    assert opt2.param_groups[0]['lr'] == 1e-3
    tr.optimizer = opt2
    m_before = opt.state[next(iter(net.parameters()))]['exp_avg'].clone()
    tr.train_step([x.to(_dev()) for x in inputs], [t.to(_dev()) for t in targets], pos.to(_dev()))
    st = opt2.state[next(iter(net.parameters()))]
    assert int(st['step']) == 4 and not torch.equal(st['exp_avg'], m_before)
    _, _, ref_grads = orc.step(cpu, cfg, [x.clone() for x in inputs], targets, pos)
    sto.adam_step([cpu[k] for k in names], [ref_grads[k] for k in names], [ms[k] for k in names], [vs[k] for k in names], 4, lr=1e-3)
    for k, p in net.named_parameters():
        assert torch.allclose(p.detach().cpu(), cpu[k], rtol=0, atol=3e-5), k


@pytest.mark.gpu
def test_flat_adam_param_groups_and_misaligned_runs():
    """Two parameter groups with their own hyper-parameters; gradients that are views of one flat buffer (the engine's
    layout) with odd sizes, so that the run after the gradient-less parameter starts off a 16-byte boundary; a
    non-contiguous gradient takes the staging copy.  Reference: torch.optim.Adam on the CPU, same inputs."""
    from hipvsr.step_tail import FlatAdam
    g = torch.Generator('cpu').manual_seed(5)
    shapes = [(3,), (5, 1), (1,), (7,), (2, 3)]
    p0 = [torch.randn(*s, generator=g) for s in shapes]
    q0 = [torch.randn(4, 6, generator=g), torch.randn(9, generator=g)]
    ref_p = [torch.nn.Parameter(p.clone()) for p in p0 + q0]
    ref = torch.optim.Adam([dict(params=ref_p[:5], lr=2e-3), dict(params=ref_p[5:], lr=5e-4, betas=(0.7, 0.9), weight_decay=0.1)])
    dev_p = [torch.nn.Parameter(p.clone().to(_dev())) for p in p0 + q0]
    opt = FlatAdam([dict(params=dev_p[:5], lr=2e-3), dict(params=dev_p[5:], lr=5e-4, betas=(0.7, 0.9), weight_decay=0.1)])
    n = sum(p.numel() for p in p0)
    for step in range(4):
        flat = torch.randn(n, generator=g)
        dflat = flat.to(_dev())
        off = 0
        for i, (rp, dp_) in enumerate(zip(ref_p[:5], dev_p[:5])):
            k = rp.numel()
            rp.grad = None if i == 2 else flat[off:off + k].view_as(rp).clone()
            dp_.grad = None if i == 2 else dflat[off:off + k].view_as(dp_)
            off += k
        gq = torch.randn(6, 4, generator=g)
        ref_p[5].grad, dev_p[5].grad = gq.t().clone(), gq.to(_dev()).t()            # non-contiguous on the device
        g9 = torch.randn(9, generator=g)
        ref_p[6].grad, dev_p[6].grad = g9.clone(), g9.to(_dev())
        ref.step()
        opt.step()
        assert opt.launches == 4                                # runs: [p0, p1], [p3, p4], [q0 (staged copy)], [q1 (its own tensor)]
        for rp, dp_ in zip(ref_p, dev_p):
            assert torch.allclose(dp_.detach().cpu(), rp.detach(), rtol=1e-6, atol=1e-8), step
    assert torch.equal(dev_p[2].detach().cpu(), p0[2])
    sd = opt.state_dict()
    assert [g_['lr'] for g_ in sd['param_groups']] == [2e-3, 5e-4] and len(sd['state']) == 6


@pytest.mark.gpu
@pytest.mark.parametrize('S,T', [(3, 7), (1, 3), (2, 1)])
def test_loss_total_vs_the_trainers_list_arithmetic(S, T):
    """rnh_loss_total / hipvsr.autograd.LossTotalFn (the discounted deep-supervision sum in one launch, its gradient in one
    more) against the reference trainer's own arithmetic (acdc_vsr_refinenet_trainer.py:83-94: per group the mean over the
    frames of loss * 0.5^(S-1-g/3), summed over the groups) evaluated by ATen in float64: value to 2 ulp-ish (rtol 1e-6),
    gradient exactly discount / T up to one rounding, upstream gradient honoured."""
    import numpy as np
    from hipvsr.autograd import LossTotalFn
    from hipvsr.hip_ops import HipOps
    dev = _dev()
    ops = HipOps(dev)
    G = 3 * S
    g = torch.Generator('cpu').manual_seed(100 * S + T)
    per = torch.rand(G * T, generator=g) * 3
    disc = [float(np.power(0.5, (G // 3 - k // 3 - 1))) for k in range(G)]
    x = per.to(dev).requires_grad_(True)
    w = torch.tensor(disc, dtype=torch.float32, device=dev)
    tot = LossTotalFn.apply(ops, x, w, G, T)
    (tot * 1.75).backward()
    xr = per.double().requires_grad_(True)
    ref = torch.stack([(xr[k * T:(k + 1) * T] * disc[k]).mean() for k in range(G)]).sum()
    (ref * 1.75).backward()
    assert tot.shape == () and abs(float(tot) - float(ref)) <= 1e-6 * abs(float(ref))
    assert torch.allclose(x.grad.cpu().double(), xr.grad, rtol=1e-6, atol=0)
    with pytest.raises(Exception):
        ops.loss_total(x.detach(), w[:-1], G, T)
