"""Metric modules of the RefineNet path (reference src/model/metrics.py): PSNR (:9-36), SSIM (:39-113), CardiacPSNR
(:116-141), CardiacSSIM (:144-169) - same constructors, same ``metric_fn(output, target[, name])`` contract - computed
by one HIP launch (``rnh_metrics_psnr_ssim``: squared error and the separable 11x11 windowed moments in one pass over
both images).  ``fused_metrics`` serves the trainer / predictor: denormalisation, PSNR and SSIM of ALL frames of a
step in one launch instead of 2 x T denormalisations and T x (MSE + five depthwise convolutions).

HIP tensors only (fp32): there is no CPU path, a CPU tensor raises ``HipKernelError``."""
import pickle

import torch
import torch.nn as nn

from hipvsr import step_tail
from hipvsr.hip_ops import packed_view

__all__ = ['PSNR', 'SSIM', 'CardiacPSNR', 'CardiacSSIM']

DENORM = {'acdc': (54.089, 48.084), 'dsb15': (51.193, 52.671)}      # reference src/utils.py:13-16


def _planes(output, target):
    """(N, C, H, W) -> (P = N*C planes, cps = C, H, W); any other (N, C, *) shape is one H=1 row per sample (PSNR only)."""
    if output.shape != target.shape:
        raise RuntimeError(f'The size of output {tuple(output.shape)} must match the size of target {tuple(target.shape)}')
    o, t = output.detach().contiguous(), target.detach().contiguous()
    if o.dim() == 4:
        n, c, h, w = o.shape
        return o, t, n * c, c, h, w
    n = o.shape[0]
    return o, t, n, 1, 1, o.numel() // max(n, 1)


class PSNR(nn.Module):
    """10 log10(max_value^2 / (mse + 1e-10)) per sample, averaged over the batch if ``size_average``."""

    def __init__(self, size_average=True, max_value=255):
        super().__init__()
        self.size_average, self.max_value = size_average, max_value

    def forward(self, output, target):
        o, t, P, cps, H, W = _planes(output, target)
        res = step_tail.psnr_ssim(o, t, P, cps, H, W, None, float(self.max_value), 255.0, want_ssim=False)
        return res[0] if self.size_average else res[2:2 + P // cps]


class SSIM(nn.Module):
    """Mean of the SSIM map over an 11x11 window (the reference's Gaussian, valid region only)."""

    def __init__(self, dim=2, channels=1, size_average=True, value_range=255):
        super().__init__()
        if dim not in (2, 3):
            raise ValueError(f'Only dim=2, 3 are supported. Received dim={dim}.')
        if dim == 3:
            raise NotImplementedError('SSIM(dim=3) is not on the cine path (frames are 2-D); only dim=2 is built')
        self.dim, self.channels, self.size_average, self.value_range = dim, channels, size_average, value_range
        self.c1, self.c2 = (0.01 * value_range) ** 2, (0.03 * value_range) ** 2
        w = torch.tensor(list(step_tail.ssim_window_1d()), dtype=torch.float32)
        # the reference registers its window as the buffer 'weight' (part of the module's state_dict)
        self.register_buffer('weight', torch.outer(w, w).view(1, 1, 11, 11).repeat(channels, 1, 1, 1))
        self.groups = channels

    def forward(self, output, target):
        if output.dim() != 4 or output.shape[1] != self.channels:
            raise RuntimeError(f'SSIM(dim=2, channels={self.channels}) expects (N, {self.channels}, H, W), got {tuple(output.shape)}')
        o, t, P, cps, H, W = _planes(output, target)
        res = step_tail.psnr_ssim(o, t, P, cps, H, W, None, 255.0, float(self.value_range))
        if self.size_average:
            return res[1]
        per_plane = res[2 + P // cps:2 + P // cps + P]
        return per_plane if cps == 1 else per_plane.view(-1, cps).mean(1)


class _Cardiac(nn.Module):
    def __init__(self, coordinates_path):
        super().__init__()
        with open(coordinates_path, 'rb') as f:
            self.coordinates = pickle.load(f)

    def _crop(self, output, target, name):
        h0, hn, w0, wn = self.coordinates[name]
        return output[..., h0:hn, w0:wn], target[..., h0:hn, w0:wn]


class CardiacPSNR(_Cardiac):
    """PSNR inside the cardiac bounding box of patient ``name``."""

    def __init__(self, coordinates_path, **kwargs):
        super().__init__(coordinates_path)
        self.psnr = PSNR(**kwargs)

    def forward(self, output, target, name):
        return self.psnr(*self._crop(output, target, name))


class CardiacSSIM(_Cardiac):
    """SSIM inside the cardiac bounding box of patient ``name``."""

    def __init__(self, coordinates_path, **kwargs):
        super().__init__(coordinates_path)
        self.ssim = SSIM(**kwargs)

    def forward(self, output, target, name):
        return self.ssim(*self._crop(output, target, name))


def fused_metrics(outputs_last, targets, metric_fns, dataset='acdc', packed_last=None, per_frame=False, per_sample=False):
    """All frames of a step in ONE launch.  per_frame=False (trainer, acdc_vsr_refinenet_trainer.py:103-120): list of
    0-dim tensors, the mean over the frames of each metric's batch-averaged score.  per_frame=True (predictor,
    acdc_vsr_refinenet_predictor.py:140-160): the (T, len(metric_fns)) tensor of per-frame scores; per_sample=True: the
    (T, N, len(metric_fns)) tensor of per-frame, per-sample scores (a predictor that runs N cines at once).  None when this
    combination is not served (then the caller does what the reference does: denormalize + metric_fn per frame).

    outputs_last: list[T] of (N, 1, H, W); packed_last: the (T*N, H, W, 1) tensor they are views of, if known."""
    fns = list(metric_fns)
    if not fns or any(type(f) not in (PSNR, SSIM) or not f.size_average for f in fns):
        return None
    ps, ss = [f for f in fns if type(f) is PSNR], [f for f in fns if type(f) is SSIM]
    if len({f.max_value for f in ps}) > 1 or len({f.value_range for f in ss}) > 1 or any(f.channels != 1 for f in ss):
        return None
    o0 = outputs_last[0]
    if not (o0.is_cuda and o0.dtype == torch.float32 and o0.dim() == 4 and o0.shape[1] == 1):
        return None
    T, (N, _, H, W) = len(outputs_last), o0.shape
    if ss and (H < 11 or W < 11):
        return None
    o = packed_last if packed_last is not None else packed_view([x.detach() for x in outputs_last])
    y = packed_view([t.detach() for t in targets])
    res = step_tail.psnr_ssim(o.contiguous(), y.contiguous(), T * N, 1, H, W, DENORM[dataset], float(ps[0].max_value) if ps else 255.0,
                              float(ss[0].value_range) if ss else 255.0, want_ssim=bool(ss))
    if not (per_frame or per_sample):
        return [res[0] if type(f) is PSNR else res[1] for f in fns]          # equal batch sizes: mean of means == overall mean
    per = {PSNR: res[2:2 + T * N].view(T, N), SSIM: res[2 + T * N:2 + 2 * T * N].view(T, N)}
    if per_sample:
        return torch.stack([per[type(f)] for f in fns], dim=2)
    return torch.stack([per[type(f)].mean(1) for f in fns], dim=1)
