"""CPU oracle for the RefineNet forward/backward hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The shipped path (the HIP kernels behind
``include/refinenet_hip.h``) never routes through it.

It is a functional, state-dict driven restatement in plain fp32 PyTorch (CPU) of

* ``RefineNet.forward``               reference ``src/model/nets/refine_net.py:61-135``
* ``_InBlock``                        reference ``src/model/nets/refine_net.py:188-192``
* ``ConvLSTMCell.forward``            reference ``src/model/nets/refine_net.py:247-267``
* ``_ConvLSTM.forward/_init_hidden``  reference ``src/model/nets/refine_net.py:304-332``
* ``_RefineBlock.forward``            reference ``src/model/nets/refine_net.py:157-185``
* ``_OutBlock``                       reference ``src/model/nets/refine_net.py:194-205``
* the deep-supervision loss of ``AcdcVSRRefineNetTrainer._compute_losses``
                                      reference ``src/runner/trainers/acdc_vsr_refinenet_trainer.py:76-101``
* ``CharbonnierLoss`` / ``HuberLoss`` reference ``src/model/losses.py:5-34``
* ``PSNR`` and ``denormalize``        reference ``src/model/metrics.py:20-36``, ``src/utils.py:1-20``

Parity status: PINNED.  The reference holds no tests or golden vectors of its own, so the
oracle is pinned against outputs of the reference itself, imported on CPU in the build container by
``tests/golden/make_golden.py`` and committed as fixtures under ``tests/golden/*.pt``;
``tests/test_oracle_golden.py`` checks this file against them (forward outputs, losses and every
parameter gradient, bit-exact).

The backward pass of the oracle is torch autograd over this forward (the reference has no hand-written
backward either: ``loss.backward()`` at ``acdc_vsr_refinenet_trainer.py:46``).
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------------------
# configuration / parameters
# --------------------------------------------------------------------------------------------------------
class Config(dict):
    """Constructor kwargs of the reference ``RefineNet`` (``refine_net.py:18-19``) with its defaults."""
    DEFAULTS = dict(in_channels=1, out_channels=1, num_features=(64, 64, 64), num_stages=1,
                    refine_window_size=5, upscale_factor=4, update_memory=False, num_updated_frames=0,
                    memory=True, positional_encoding=False)

    def __init__(self, **kw):
        unknown = set(kw) - set(self.DEFAULTS)
        if unknown:
            raise TypeError(f'unknown RefineNet kwargs: {sorted(unknown)}')
        merged = dict(self.DEFAULTS)
        merged.update(kw)
        merged['num_features'] = list(merged['num_features'])
        super().__init__(merged)
        # error behaviour of refine_net.py:30-34
        if self['upscale_factor'] not in (2, 3, 4, 8):
            raise ValueError(f"The upscale factor should be 2, 3, 4 or 8. Got {self['upscale_factor']}.")
        if (not self['update_memory']) and self['num_updated_frames'] != 0:
            raise ValueError('The "update_memory" is not activated!')

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def exp1_x4_config(**over):
    """net.kwargs of configs/train/refine_net/exp1_x4.yaml:35-46."""
    kw = dict(in_channels=1, out_channels=1, num_features=[64, 64, 64], upscale_factor=4, num_stages=3,
              update_memory=True, num_updated_frames=6, refine_window_size=5, positional_encoding=True)
    kw.update(over)
    return Config(**kw)


def out_block_layout(cfg):
    """[(name, cout, cin, pixel_shuffle_factor_after)] of _OutBlock (refine_net.py:194-205)."""
    c, s = cfg.num_features[0], cfg.upscale_factor
    if s == 3:
        return [('conv1', 9 * c, c, 3), ('conv2', cfg.out_channels, c, 1)]
    n = int(round(math.log2(s)))
    layers = [(f'conv{i + 1}', 4 * c, c, 2) for i in range(n)]
    layers.append((f'conv{n + 1}', cfg.out_channels, c, 1))
    return layers


def state_dict_spec(cfg):
    """Ordered name -> shape map, identical in names, shapes and order to the reference ``state_dict()``."""
    nf = cfg.num_features
    c0, cl = nf[0], nf[-1]
    spec = OrderedDict()
    spec['in_block.conv.weight'] = (c0, cfg.in_channels, 3, 3)
    spec['in_block.conv.bias'] = (c0,)
    spec['in_block.prelu.weight'] = (1,)
    for d in ('forward', 'backward'):
        for i, hd in enumerate(nf):
            cin = (c0 if i == 0 else nf[i - 1])
            cin = cin + hd if cfg.memory else 2 * cin
            spec[f'{d}_lstm_block.cell_list.{i}.conv.weight'] = (4 * hd, cin, 3, 3)
            spec[f'{d}_lstm_block.cell_list.{i}.conv.bias'] = (4 * hd,)
    w = cfg.refine_window_size
    if cfg.positional_encoding:
        rin = w * (2 * cl + 1)
        spec['refine_block.body.conv1.weight'] = (rin // w, rin, 3, 3)
        spec['refine_block.body.conv1.bias'] = (rin // w,)
        spec['refine_block.body.conv2.weight'] = (cl, rin // w, 3, 3)
        spec['refine_block.body.conv2.bias'] = (cl,)
    else:
        rin = w * 2 * cl
        spec['refine_block.body.conv1.weight'] = (cl, rin, 1, 1)
        spec['refine_block.body.conv1.bias'] = (cl,)
    spec['refine_block.prelu.weight'] = (1,)           # registered but never applied (refine_net.py:150-155)
    for name, cout, cin, _ in out_block_layout(cfg):
        spec[f'out_block.{name}.weight'] = (cout, cin, 3, 3)
        spec[f'out_block.{name}.bias'] = (cout,)
    return spec


def init_state_dict(cfg, seed=0):
    """Deterministic parameters with the statistics of the nn.Conv2d / nn.PReLU(init=0.2) defaults.

    Conv weight and bias ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (what kaiming_uniform_(a=sqrt(5)) reduces to),
    drawn from a CPU generator in state-dict order, so the same seed gives the same tensors everywhere.
    """
    g = torch.Generator('cpu').manual_seed(int(seed))
    sd = OrderedDict()
    spec = state_dict_spec(cfg)
    for name, shape in spec.items():
        if name.endswith('prelu.weight'):
            sd[name] = torch.full(shape, 0.2, dtype=torch.float32)
            continue
        wname = name.rsplit('.', 1)[0] + '.weight'
        wshape = spec[wname]
        bound = 1.0 / math.sqrt(wshape[1] * wshape[2] * wshape[3])
        sd[name] = (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * bound
    return sd


def as_leaf_params(sd):
    """Clone a state dict into fp32 leaf tensors that require grad (for autograd in the oracle)."""
    return OrderedDict((k, v.detach().clone().float().requires_grad_(True)) for k, v in sd.items())


# --------------------------------------------------------------------------------------------------------
# blocks
# --------------------------------------------------------------------------------------------------------
def in_block(p, x):
    """refine_net.py:188-192: PReLU_a(conv3x3(x) + b)."""
    y = F.conv2d(x, p['in_block.conv.weight'], p['in_block.conv.bias'], padding=1)
    return F.prelu(y, p['in_block.prelu.weight'])


def lstm_cell(p, prefix, x, h, c, memory=True):
    """refine_net.py:247-267; gate order i, f, o, g (:258)."""
    combined = torch.cat([x, h], dim=1) if memory else torch.cat([x, x], dim=1)
    cc = F.conv2d(combined, p[prefix + '.conv.weight'], p[prefix + '.conv.bias'], padding=1)
    hd = cc.shape[1] // 4
    cc_i, cc_f, cc_o, cc_g = torch.split(cc, hd, dim=1)
    i = torch.sigmoid(cc_i)
    f = torch.sigmoid(cc_f)
    o = torch.sigmoid(cc_o)
    g = torch.tanh(cc_g)
    c_next = f * c + i * g
    h_next = o * torch.tanh(c_next)
    return h_next, c_next


class _LSTMState:
    """Stateful multi-layer ConvLSTM (refine_net.py:304-332): zero state, then one call per frame."""

    def __init__(self, p, direction, cfg, n, hgt, wid):
        self.p, self.cfg = p, cfg
        self.prefix = f'{direction}_lstm_block.cell_list'
        self.state = [(torch.zeros(n, hd, hgt, wid), torch.zeros(n, hd, hgt, wid)) for hd in cfg.num_features]

    def step(self, x):
        cur = x
        for li in range(len(self.cfg.num_features)):
            h, c = self.state[li]
            h, c = lstm_cell(self.p, f'{self.prefix}.{li}', cur, h, c, self.cfg.memory)
            self.state[li] = (h, c)
            cur = h
        return cur


def refine_body(p, cfg, x):
    """_RefineBlock.body (refine_net.py:147-155).  No activation is applied anywhere (quirk Q1)."""
    if cfg.positional_encoding:
        y = F.conv2d(x, p['refine_block.body.conv1.weight'], p['refine_block.body.conv1.bias'], padding=1)
        return F.conv2d(y, p['refine_block.body.conv2.weight'], p['refine_block.body.conv2.bias'], padding=1)
    return F.conv2d(x, p['refine_block.body.conv1.weight'], p['refine_block.body.conv1.bias'])


def refine_block(p, cfg, hf, hb, pos_codes):
    """refine_net.py:157-185."""
    n, c, hgt, wid = hf[0].shape
    hw = cfg.refine_window_size // 2
    U = cfg.num_updated_frames
    nfr = len(hf)
    ff = torch.stack(hf, dim=1)
    bb = torch.stack(hb, dim=1)
    pc = pos_codes.repeat(hgt, wid, 1, 1, 1).permute(2, 3, 4, 0, 1).contiguous()      # (N, F, 1, H, W)
    feats = torch.cat((ff, bb, pc), dim=2) if cfg.positional_encoding else torch.cat((ff, bb), dim=2)
    maps = []
    for k in range(hw, nfr - hw):
        win = feats[:, k - hw:k + hw + 1]
        win = torch.cat([win[:, j] for j in range(win.shape[1])], dim=1)
        if U <= k < nfr - U:
            maps.append(refine_body(p, cfg, win))
        else:
            with torch.no_grad():
                maps.append(refine_body(p, cfg, win))
    return maps


def out_block(p, cfg, x):
    """refine_net.py:194-205: [conv, PixelShuffle]* then conv; affine, no activation."""
    for name, _, _, r in out_block_layout(cfg):
        x = F.conv2d(x, p[f'out_block.{name}.weight'], p[f'out_block.{name}.bias'], padding=1)
        if r > 1:
            x = F.pixel_shuffle(x, r)
    return x


# --------------------------------------------------------------------------------------------------------
# the network
# --------------------------------------------------------------------------------------------------------
def forward(p, cfg, inputs, pos_codes):
    """RefineNet.forward (refine_net.py:61-135).

    inputs: list[F] of (N, in_ch, H, W); pos_codes: (N, F, 1).
    Returns tuple[3*S] of list[T] of (N, out_ch, sH, sW); per stage the order is [forward, backward, fused].
    """
    U, S, hw = cfg.num_updated_frames, cfg.num_stages, cfg.refine_window_size // 2
    nfr = len(inputs)
    centre = inputs[U:-U]            # U == 0 gives an empty list, like the reference (quirk Q2)
    T = nfr - 2 * U
    feat_c = [in_block(p, x) for x in centre]
    feat_f, feat_b = [], []
    outputs = []
    for _ in range(S):
        n, _, hgt, wid = feat_c[0].shape            # IndexError for U == 0, as refine_net.py:71
        lf = _LSTMState(p, 'forward', cfg, n, hgt, wid)
        lb = _LSTMState(p, 'backward', cfg, n, hgt, wid)
        with torch.no_grad():
            if len(feat_f) == 0:
                feat_f = [in_block(p, x) for x in inputs[:U]]
                feat_b = [in_block(p, x) for x in inputs[-U:]]
        feats = feat_f + feat_c + feat_b
        nall = len(feats)
        hf, hb = [], []
        for i, ft in enumerate(feats):
            if U <= i < nall - U:
                hf.append(lf.step(ft))
            else:
                with torch.no_grad():
                    hf.append(lf.step(ft))
        for i, ft in enumerate(reversed(feats)):
            if U <= i < nall - U:
                hb.insert(0, lb.step(ft))
            else:
                with torch.no_grad():
                    hb.insert(0, lb.step(ft))
        R = refine_block(p, cfg, hf, hb, pos_codes)

        outputs.append([out_block(p, cfg, feat_c[i] + hf[i + U]) for i in range(T)])
        outputs.append([out_block(p, cfg, feat_c[i] + hb[i + U]) for i in range(T)])
        outputs.append([out_block(p, cfg, feat_c[i] + R[i + U - hw]) for i in range(T)])

        if S > 1:                                   # refine_net.py:118-133 (in-place updates)
            for i in range(len(feat_f)):
                if i < hw:
                    feat_f[i] += hf[i]
                else:
                    feat_f[i] += R[i - hw]
            for i in range(len(feat_b)):
                if i < hw:
                    feat_b[-i - 1] += hb[-i - 1]
                else:
                    feat_b[-i - 1] += R[-i + hw - 1]
            for i in range(len(feat_c)):
                feat_c[i] += R[i + U - hw]
    return tuple(outputs)


# --------------------------------------------------------------------------------------------------------
# losses / metrics
# --------------------------------------------------------------------------------------------------------
def l1_loss(output, target):
    return F.l1_loss(output, target)


def charbonnier_loss(output, target, epsilon=1e-6):
    """src/model/losses.py:32-34."""
    return torch.mean(torch.sqrt((output - target) ** 2 + epsilon))


def huber_loss(output, target, delta):
    """src/model/losses.py:14-20."""
    abs_error = torch.abs(output - target)
    d = torch.ones_like(output) * delta
    quadratic = torch.min(abs_error, d)
    linear = abs_error - quadratic
    return torch.mean(0.5 * quadratic ** 2 + d * linear)


def training_loss(outputs, targets, loss_fn=l1_loss):
    """Training branch of _compute_losses (acdc_vsr_refinenet_trainer.py:83-94) for one loss function."""
    terms = []
    for g, group in enumerate(outputs):
        discount = np.power(0.5, (len(outputs) // 3 - g // 3 - 1))
        terms.append(torch.stack([loss_fn(o, t) * discount for o, t in zip(group, targets)]).mean())
    return torch.stack(terms).sum()


def eval_loss(outputs, targets, loss_fn=l1_loss):
    """Evaluation branch (acdc_vsr_refinenet_trainer.py:95-100): last group only."""
    return torch.stack([loss_fn(o, t) for o, t in zip(outputs[-1], targets)]).mean()


def denormalize(imgs, dataset='acdc'):
    """src/utils.py:1-20."""
    if dataset not in ('acdc', 'dsb15'):
        raise ValueError(f"The name of the dataset should be 'acdc' or 'dsb15'. Got {dataset}.")
    mean, std = (54.089, 48.084) if dataset == 'acdc' else (51.193, 52.671)
    return (imgs.clone() * std + mean).round().clamp(0, 255)


def psnr(output, target, max_value=255, size_average=True):
    """src/model/metrics.py:20-36."""
    dims = list(range(1, output.dim()))
    mse = F.mse_loss(output, target, reduction='none').mean(dims)
    val = 10 * torch.log10(max_value ** 2 / (mse + 1e-10))
    return val.mean() if size_average else val


def frame_psnr(outputs_last, targets):
    """PSNR part of _compute_metrics (acdc_vsr_refinenet_trainer.py:103-120): per-frame values."""
    return [psnr(denormalize(o), denormalize(t)) for o, t in zip(outputs_last, targets)]


# --------------------------------------------------------------------------------------------------------
# synthetic inputs and a complete step
# --------------------------------------------------------------------------------------------------------
def synthetic_batch(cfg, n, t, h, w, seed, dtype=torch.float32):
    """'synthetic Gaussian cine': x_k, Y_i ~ N(0,1); p_{n,k} = cos(2 pi (k + phi_n) / 30), phi_n ~ U{0..29}."""
    g = torch.Generator('cpu').manual_seed(int(seed))
    nfr = t + 2 * cfg.num_updated_frames
    s = cfg.upscale_factor
    inputs = [torch.randn(n, cfg.in_channels, h, w, generator=g, dtype=dtype) for _ in range(nfr)]
    targets = [torch.randn(n, cfg.out_channels, s * h, s * w, generator=g, dtype=dtype) for _ in range(t)]
    phi = torch.randint(0, 30, (n, 1), generator=g).to(dtype)
    k = torch.arange(nfr, dtype=dtype).unsqueeze(0)
    pos = torch.cos(2 * math.pi * (k + phi) / 30.0).unsqueeze(-1)          # (N, F, 1)
    return inputs, targets, pos



def structured_cine(cfg, n, t, h, w, seed, dtype=torch.float32):
    """The parity / PSNR inputs of SURVEY.md section 8(d): an HR cine = Gaussian-blurred (sigma = 3 px) noise field, slowly
    rotating between two such fields over the cycle, plus a disc whose radius varies sinusoidally in time (the "ventricle"),
    scaled into [0, 255]; LR = avg_pool(scale) of HR; both normalised like the reference's ACDC config ((x - 54.089) / 48.084,
    configs/train/refine_net/exp1_x4.yaml:12-15); phase codes as in synthetic_batch.  Returns (inputs[F], targets[T], pos)."""
    g = torch.Generator('cpu').manual_seed(int(seed))
    U, s = cfg.num_updated_frames, cfg.upscale_factor
    nfr, Hh, Wh = t + 2 * U, s * h, s * w
    r = torch.arange(-9, 10, dtype=torch.float64)
    k1 = torch.exp(-r * r / (2 * 3.0 ** 2))
    k1 = (k1 / k1.sum()).view(1, 1, 1, -1)

    def blurred():
        z = torch.randn(n, 1, Hh, Wh, generator=g, dtype=torch.float64)
        z = F.conv2d(F.pad(z, (9, 9, 0, 0), mode='reflect'), k1)
        z = F.conv2d(F.pad(z, (0, 0, 9, 9), mode='reflect'), k1.transpose(2, 3))
        return (z - z.mean(dim=(2, 3), keepdim=True)) / z.std(dim=(2, 3), keepdim=True)

    f0, f1 = blurred(), blurred()
    phi = torch.randint(0, 30, (n, 1), generator=g)
    cy = (0.35 + 0.3 * torch.rand(n, generator=g, dtype=torch.float64)) * Hh
    cx = (0.35 + 0.3 * torch.rand(n, generator=g, dtype=torch.float64)) * Wh
    r0 = 0.18 * min(Hh, Wh)
    yy = torch.arange(Hh, dtype=torch.float64).view(1, Hh, 1)
    xx = torch.arange(Wh, dtype=torch.float64).view(1, 1, Wh)
    dist = torch.sqrt((yy - cy.view(n, 1, 1)) ** 2 + (xx - cx.view(n, 1, 1)) ** 2)
    hr = []
    for k in range(nfr):
        th = 2 * math.pi * (k + phi.to(torch.float64)) / 30.0                                  # (n, 1)
        field = torch.cos(0.5 * th).view(n, 1, 1, 1) * f0 + torch.sin(0.5 * th).view(n, 1, 1, 1) * f1
        rad = r0 * (1.0 + 0.25 * torch.sin(th)).view(n, 1, 1)
        disc = torch.sigmoid((rad - dist) / 1.5).unsqueeze(1)
        hr.append((100.0 + 35.0 * field + 90.0 * disc).clamp(0, 255))
    norm = lambda x: ((x - 54.089) / 48.084).to(dtype)                                        # noqa: E731
    inputs = [norm(F.avg_pool2d(x, s)) for x in hr]
    targets = [norm(x) for x in hr[U:U + t]]
    kk = torch.arange(nfr, dtype=dtype).unsqueeze(0)
    pos = torch.cos(2 * math.pi * (kk + phi.to(dtype)) / 30.0).unsqueeze(-1)
    return inputs, targets, pos


def step(sd, cfg, inputs, targets, pos_codes, loss_fn=l1_loss):
    """One forward + training loss + backward on CPU.  Returns (outputs, loss, grads dict; None where unused)."""
    p = as_leaf_params(sd)
    outs = forward(p, cfg, inputs, pos_codes)
    loss = training_loss(outs, targets, loss_fn)
    loss.backward()
    grads = OrderedDict((k, (v.grad.detach().clone() if v.grad is not None else None)) for k, v in p.items())
    outs = tuple([o.detach() for o in grp] for grp in outs)
    return outs, loss.detach(), grads
