"""Golden-vector generator.  Runs ONLY in the build container, where /root/reference is mounted.

It imports the reference's own Python (``RefineNet``, the trainer's loss schedule, the losses, PSNR and
``denormalize``) on CPU, by file path, with the two shims documented in SURVEY.md §8(c):

  1. empty stub packages for ``src`` / ``src.model`` / ``src.model.nets`` / ``src.runner`` /
     ``src.runner.trainers`` so that ``src/__init__.py`` (which pulls in nibabel, SimpleITK, box, ...) is
     bypassed;
  2. ``ConvLSTMCell.init_hidden`` without its hard-coded ``.cuda()`` (refine_net.py:269-271).

Nothing of the reference is copied: the outputs written here are data (inputs, weights, expected
outputs / losses / gradients).  Usage:  python tests/golden/make_golden.py
"""
import functools
import importlib.util
import itertools
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference/src/'
HERE = os.path.dirname(os.path.abspath(__file__))


def _load_reference():
    for n in ['src', 'src.model', 'src.model.nets', 'src.runner', 'src.runner.trainers']:
        m = types.ModuleType(n)
        m.__path__ = []
        sys.modules[n] = m

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, REF + rel)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    mods = types.SimpleNamespace()
    mods.utils = load('src.utils', 'utils.py')
    load('src.model.nets.base_net', 'model/nets/base_net.py')
    mods.rn = load('src.model.nets.refine_net', 'model/nets/refine_net.py')
    mods.rn.ConvLSTMCell.init_hidden = lambda self, b, h, w: (torch.zeros(b, self.hidden_dim, h, w),
                                                              torch.zeros(b, self.hidden_dim, h, w))
    mods.losses = load('src.model.losses', 'model/losses.py')
    mods.metrics = load('src.model.metrics', 'model/metrics.py')
    load('src.runner.trainers.base_trainer', 'runner/trainers/base_trainer.py')
    mods.trainer = load('src.runner.trainers.acdc_vsr_refinenet_trainer',
                        'runner/trainers/acdc_vsr_refinenet_trainer.py')
    return mods


def _bare_trainer(mods, net, loss_fns, metric_fns):
    tr = object.__new__(mods.trainer.AcdcVSRRefineNetTrainer)
    tr.net = net
    tr.loss_fns = loss_fns
    tr.metric_fns = metric_fns
    tr._denormalize = functools.partial(mods.utils.denormalize, dataset='acdc')
    return tr


def _make_inputs(g, n, nfr, t, cin, cout, h, w, s):
    inputs = [torch.randn(n, cin, h, w, generator=g) for _ in range(nfr)]
    targets = [torch.randn(n, cout, s * h, s * w, generator=g) for _ in range(t)]
    pos = torch.rand(n, nfr, 1, generator=g) * 2 - 1
    return inputs, targets, pos


def g1_tiny(mods):
    """Tiny nets, every output / loss / gradient kept in full."""
    cases = {}
    combos = list(itertools.product([2, 3, 4], [True, False], [True, False])) + [(8, True, True)]
    for s, pos_enc, memory in combos:
        torch.manual_seed(1000 + s * 10 + int(pos_enc) * 2 + int(memory))
        kwargs = dict(in_channels=1, out_channels=1, num_features=[8, 8], num_stages=3, refine_window_size=5,
                      upscale_factor=s, update_memory=True, num_updated_frames=3, memory=memory,
                      positional_encoding=pos_enc)
        net = mods.rn.RefineNet(**kwargs)
        g = torch.Generator('cpu').manual_seed(77 + s)
        n, t, h, w = 2, 4, 6, 5
        inputs, targets, pos = _make_inputs(g, n, t + 6, t, 1, 1, h, w, s)
        sd = {k: v.detach().clone() for k, v in net.state_dict().items()}

        net.train()
        tr = _bare_trainer(mods, net, [torch.nn.L1Loss()], [])
        outs = net([x.clone() for x in inputs], pos.clone())
        loss = tr._compute_losses(outs, targets)[0]
        net.zero_grad()
        loss.backward()
        grads = {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in net.named_parameters()}

        net.eval()
        with torch.no_grad():
            outs_e = net([x.clone() for x in inputs], pos.clone())
            loss_e = tr._compute_losses(outs_e, targets)[0]

        # Charbonnier (the other loss north_star names) through the same schedule
        net.train()
        tr_c = _bare_trainer(mods, net, [mods.losses.CharbonnierLoss(epsilon=1e-6)], [])
        outs_c = net([x.clone() for x in inputs], pos.clone())
        loss_c = tr_c._compute_losses(outs_c, targets)[0]
        net.zero_grad()
        loss_c.backward()
        grads_c = {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in net.named_parameters()}

        cases[f'x{s}_pos{int(pos_enc)}_mem{int(memory)}'] = dict(
            kwargs=kwargs, state_dict=sd, inputs=inputs, targets=targets, pos_codes=pos,
            outputs=[[o.detach().clone() for o in grp] for grp in outs],
            train_loss=loss.detach().clone(), grads=grads,
            eval_loss=loss_e.detach().clone(),
            eval_last=[o.clone() for o in outs_e[-1]],
            charbonnier_train_loss=loss_c.detach().clone(), charbonnier_grads=grads_c)
    torch.save(cases, os.path.join(HERE, 'g1_tiny.pt'))
    print('g1_tiny.pt', len(cases), 'cases')


def g2_cfg1(mods):
    """Full-width BASELINE config 1 (x4, N=1, T=3, 64x64): digests only, inputs regenerated from seeds."""
    sys.path.insert(0, os.path.join(HERE, '..', '..'))
    from oracle import refinenet_oracle as orc
    cfg = orc.exp1_x4_config()
    sd = orc.init_state_dict(cfg, seed=20200526)
    net = mods.rn.RefineNet(**{k: cfg[k] for k in cfg})
    net.load_state_dict(sd)
    inputs, targets, pos = orc.synthetic_batch(cfg, n=1, t=3, h=64, w=64, seed=20200526 + 1)
    net.train()
    tr = _bare_trainer(mods, net, [torch.nn.L1Loss()], [mods.metrics.PSNR()])
    outs = net([x.clone() for x in inputs], pos.clone())
    loss = tr._compute_losses(outs, targets)[0]
    net.zero_grad()
    loss.backward()
    psnr = tr._compute_metrics(outs, targets)[0]
    rec = dict(
        seed_weights=20200526, seed_inputs=20200526 + 1, n=1, t=3, h=64, w=64,
        train_loss=float(loss.detach().double()),
        psnr=float(psnr),
        out_sum=[[float(o.detach().double().sum()) for o in grp] for grp in outs],
        out_abs_sum=[[float(o.detach().double().abs().sum()) for o in grp] for grp in outs],
        out_crop=[[o.detach()[0, 0, 100:116, 100:116].clone() for o in grp] for grp in outs],
        grad_l2={k: (float(p.grad.double().norm()) if p.grad is not None else None) for k, p in net.named_parameters()},
        grad_head={k: (p.grad.flatten()[:16].clone() if p.grad is not None else None)
                   for k, p in net.named_parameters()},
    )
    torch.save(rec, os.path.join(HERE, 'g2_cfg1.pt'))
    print('g2_cfg1.pt loss', rec['train_loss'], 'psnr', rec['psnr'])


def g3_trainer_g4_losses(mods):
    """Trainer loss schedule / metrics and the loss functions on fixed tensors."""
    g = torch.Generator('cpu').manual_seed(4242)
    outs = tuple([torch.randn(2, 1, 8, 12, generator=g) for _ in range(4)] for _ in range(9))
    targets = [torch.randn(2, 1, 8, 12, generator=g) for _ in range(4)]

    class _N:       # stand-in exposing .training like an nn.Module
        training = True
    net = _N()
    rec = dict(outputs=outs, targets=targets)
    for name, fn in [('L1Loss', torch.nn.L1Loss()), ('CharbonnierLoss', mods.losses.CharbonnierLoss(epsilon=1e-6)),
                     ('HuberLoss', mods.losses.HuberLoss(delta=0.01))]:
        tr = _bare_trainer(mods, net, [fn], [mods.metrics.PSNR()])
        net.training = True
        rec[f'{name}_train'] = tr._compute_losses(outs, targets)[0].clone()
        net.training = False
        rec[f'{name}_eval'] = tr._compute_losses(outs, targets)[0].clone()
        o = outs[0][0].clone().requires_grad_(True)
        v = fn(o, targets[0])
        v.backward()
        rec[f'{name}_value'] = v.detach().clone()
        rec[f'{name}_grad'] = o.grad.clone()
    tr = _bare_trainer(mods, net, [], [mods.metrics.PSNR()])
    rec['PSNR_metric'] = tr._compute_metrics(outs, targets)[0].clone()
    x = torch.randn(2, 1, 8, 12, generator=g) * 2
    rec['denorm_in'] = x
    rec['denorm_acdc'] = mods.utils.denormalize(x, 'acdc')
    rec['denorm_dsb15'] = mods.utils.denormalize(x, 'dsb15')
    rec['psnr_pair'] = mods.metrics.PSNR()(mods.utils.denormalize(outs[0][0], 'acdc'),
                                           mods.utils.denormalize(targets[0], 'acdc')).clone()
    torch.save(rec, os.path.join(HERE, 'g3_g4_losses.pt'))
    print('g3_g4_losses.pt')


def g5_edges(mods):
    """Constructor / shape error behaviour and a long whole-cycle evaluation (F = 30 + 12)."""
    rec = {}

    def err(fn):
        try:
            fn()
        except Exception as e:       # noqa: BLE001 - the type and message are the data recorded
            return type(e).__name__, str(e)
        return None

    base = dict(in_channels=1, out_channels=1, num_features=[8, 8])
    rec['bad_upscale'] = err(lambda: mods.rn.RefineNet(upscale_factor=5, **base))
    rec['update_memory_off'] = err(lambda: mods.rn.RefineNet(num_updated_frames=2, update_memory=False, **base))
    net0 = mods.rn.RefineNet(num_stages=2, update_memory=True, num_updated_frames=0, positional_encoding=True, **base)
    xs = [torch.zeros(1, 1, 4, 4) for _ in range(6)]
    rec['U0_forward'] = err(lambda: net0(xs, torch.zeros(1, 6, 1)))
    net1 = mods.rn.RefineNet(num_stages=2, update_memory=True, num_updated_frames=1, positional_encoding=True, **base)
    rec['U1_forward'] = err(lambda: net1(xs, torch.zeros(1, 6, 1)))

    torch.manual_seed(5)
    kwargs = dict(in_channels=1, out_channels=1, num_features=[8, 8, 8], num_stages=3, refine_window_size=5,
                  upscale_factor=4, update_memory=True, num_updated_frames=6, positional_encoding=True)
    net = mods.rn.RefineNet(**kwargs)
    net.eval()
    g = torch.Generator('cpu').manual_seed(55)
    inputs, _, pos = _make_inputs(g, 1, 42, 30, 1, 1, 7, 9, 4)
    with torch.no_grad():
        last = net([x.clone() for x in inputs], pos.clone())[-1]
    rec['cycle'] = dict(kwargs=kwargs, state_dict={k: v.clone() for k, v in net.state_dict().items()},
                        inputs=inputs, pos_codes=pos, last=[o.clone() for o in last])
    rec['param_count_exp1_x4'] = sum(p.numel() for p in mods.rn.RefineNet(
        in_channels=1, out_channels=1, num_features=[64, 64, 64], upscale_factor=4, num_stages=3, update_memory=True,
        num_updated_frames=6, refine_window_size=5, positional_encoding=True).parameters())
    torch.save(rec, os.path.join(HERE, 'g5_edges.pt'))
    print('g5_edges.pt', {k: v for k, v in rec.items() if k not in ('cycle',)})


def g6_input():
    """Input feeding (row f1): the reference's own RandomHorizontalFlip / RandomVerticalFlip / RandomCropPatch /
    Normalize / ToTensor (src/data/transforms.py), composed as in exp1_x4.yaml:11-23, on seeded synthetic cines.
    transforms.py imports SimpleITK at module level (used by RandomElasticDeformation only, which is not exercised):
    an empty placeholder module lets that import line pass."""
    import random
    sys.modules.setdefault('SimpleITK', types.ModuleType('SimpleITK'))
    for n in ['src.data']:
        m = types.ModuleType(n)
        m.__path__ = []
        sys.modules[n] = m
    spec = importlib.util.spec_from_file_location('src.data.transforms', REF + 'data/transforms.py')
    tr = importlib.util.module_from_spec(spec)
    sys.modules['src.data.transforms'] = tr
    sys.modules['src.data'].transforms = tr
    sys.modules['src'].data = sys.modules['src.data']
    spec.loader.exec_module(tr)

    class Box(dict):
        __getattr__ = dict.get

    rng = np.random.RandomState(20200526)
    cases = []
    for ci, (s, Hl, Wl, Tc, size) in enumerate([(4, 12, 15, 5, (8, 8)), (2, 10, 9, 4, (6, 7)), (3, 9, 8, 6, (9, 8)), (4, 8, 8, 3, (5, 6))]):
        hr = np.round(rng.rand(Hl * s, Wl * s, 1, Tc) * 255).astype(np.float32)
        lr = (rng.rand(Hl, Wl, 1, Tc) * 255).astype(np.float32)
        augments = tr.compose([Box(name='RandomHorizontalFlip'), Box(name='RandomVerticalFlip'),
                               Box(name='RandomCropPatch', kwargs=dict(size=list(size), ratio=s))])
        transforms = tr.compose([Box(name='Normalize', kwargs=dict(means=[54.089], stds=[48.084])), Box(name='ToTensor')])
        code = np.cos(np.linspace(0, np.pi, Tc, endpoint=False))
        runs = []
        for seed in range(6):
            imgs = [lr[..., t] for t in range(Tc)] + [hr[..., t] for t in range(Tc)]
            random.seed(1000 * ci + seed)
            aug = augments(*imgs)
            out = transforms(*aug)
            out = [o.permute(2, 0, 1).contiguous() for o in out]                      # dataset :62
            runs.append(dict(seed=1000 * ci + seed, aug=[torch.from_numpy(np.ascontiguousarray(a)) for a in aug], out=out))
        whole = [o.permute(2, 0, 1).contiguous() for o in transforms(*[lr[..., t] for t in range(Tc)], *[hr[..., t] for t in range(Tc)])]
        cases.append(dict(s=s, size=size, lr=torch.from_numpy(lr), hr=torch.from_numpy(hr), code=torch.from_numpy(code),
                          pos_code=transforms(code, normalize_tags=[False]), runs=runs, whole=whole))      # dataset :71
    torch.save(cases, os.path.join(HERE, 'g6_input.pt'))
    print('g6_input.pt', len(cases), 'cases,', os.path.getsize(os.path.join(HERE, 'g6_input.pt')), 'bytes')


def g7_metrics_adam(mods):
    """Rows f3 / f4: the reference's SSIM / PSNR / denormalize / trainer._compute_metrics on fixed tensors (random,
    smooth, image sizes that straddle the kernel's tile borders, the smallest legal 11x11 image), and the trajectory of
    torch.optim.Adam - the optimizer the reference's YAML instantiates (src/main.py:76) - on fixed gradients."""
    import torch.nn.functional as F
    g = torch.Generator('cpu').manual_seed(777)
    cases = {}

    def smooth(n, h, w):
        x = torch.randn(n, 1, h + 16, w + 16, generator=g)
        k = torch.ones(1, 1, 9, 9) / 81.0
        x = F.conv2d(F.conv2d(x, k), k)
        return (x / x.std()) * 0.8

    shapes = dict(random=(2, 24, 29, 3), smooth=(2, 40, 37, 3), wide=(1, 35, 150, 2), minimal=(3, 11, 11, 1), tall=(1, 70, 13, 2))
    for name, (n, h, w, t) in shapes.items():
        if name == 'random':
            outs = [torch.randn(n, 1, h, w, generator=g) for _ in range(t)]
            tgts = [torch.randn(n, 1, h, w, generator=g) for _ in range(t)]
        else:
            tgts = [smooth(n, h, w) for _ in range(t)]
            outs = [y + 0.05 * torch.randn(n, 1, h, w, generator=g) for y in tgts]
        outputs = tuple([torch.zeros_like(o) for o in outs] for _ in range(8)) + (outs,)
        tr = _bare_trainer(mods, None, [], [mods.metrics.PSNR(), mods.metrics.SSIM()])
        m = tr._compute_metrics(outputs, tgts)
        den_o = [mods.utils.denormalize(o, 'acdc') for o in outs]
        den_t = [mods.utils.denormalize(y, 'acdc') for y in tgts]
        cases[name] = dict(outputs_last=outs, targets=tgts, trainer_psnr=m[0].clone(), trainer_ssim=m[1].clone(),
                           psnr_per_sample=torch.stack([mods.metrics.PSNR(size_average=False)(a, b) for a, b in zip(den_o, den_t)]),
                           ssim_per_sample=torch.stack([mods.metrics.SSIM(size_average=False)(a, b) for a, b in zip(den_o, den_t)]))
    o, y = torch.rand(2, 1, 20, 23, generator=g), torch.rand(2, 1, 20, 23, generator=g)
    cases['unit_range'] = dict(output=o, target=y, psnr=mods.metrics.PSNR(max_value=1)(o, y).clone(),
                               ssim=mods.metrics.SSIM(value_range=1)(o, y).clone())
    cases['ssim_weight'] = mods.metrics.SSIM().weight.clone()

    adam = {}
    for name, kw in dict(yaml=dict(lr=1e-4, weight_decay=0), decay=dict(lr=3e-3, betas=(0.8, 0.99), eps=1e-6, weight_decay=0.01)).items():
        shapes_p = [(7, 3, 3, 3), (7,), (1,), (5, 4)]
        p0 = [torch.randn(*s_, generator=g) * 0.1 for s_ in shapes_p]
        grads = [[(torch.randn(*s_, generator=g) * (10.0 ** (i - 2)) if (k, i) != (1, 2) else torch.zeros(*s_)) for i, s_ in enumerate(shapes_p)]
                 for k in range(6)]
        ps = [torch.nn.Parameter(p.clone()) for p in p0]
        opt = torch.optim.Adam(ps[:2] + ps[2:], **kw)
        traj = []
        for gs in grads:
            for p_, g_ in zip(ps, gs):
                p_.grad = g_.clone()
            ps[2].grad = None                       # a parameter that never receives a gradient (quirk Q1)
            opt.step()
            traj.append([p_.detach().clone() for p_ in ps])
        adam[name] = dict(kwargs=kw, p0=p0, grads=grads, trajectory=traj)
    cases['adam'] = adam
    torch.save(cases, os.path.join(HERE, 'g7_metrics.pt'))
    print('g7_metrics.pt', {k: (float(v['trainer_psnr']), float(v['trainer_ssim'])) for k, v in cases.items() if isinstance(v, dict) and 'trainer_psnr' in v})


if __name__ == '__main__':
    torch.set_num_threads(8)
    mods = _load_reference()
    if sys.argv[1:] == ['g6']:
        g6_input()
        sys.exit(0)
    if sys.argv[1:] == ['g7']:
        g7_metrics_adam(mods)
        sys.exit(0)
    g1_tiny(mods)
    g3_trainer_g4_losses(mods)
    g5_edges(mods)
    g2_cfg1(mods)
    g6_input()
    g7_metrics_adam(mods)
