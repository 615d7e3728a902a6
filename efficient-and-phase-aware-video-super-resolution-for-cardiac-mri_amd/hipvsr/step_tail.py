"""What follows the backward pass in a training step, on the device:

* ``FlatAdam``: the optimizer the reference's YAML asks for (``optimizer: {name: Adam, kwargs: {lr, weight_decay}}``,
  reference ``src/main.py:76``, ``configs/train/refine_net/exp1_x4.yaml:56-60``) with parameters, gradients and both
  moments each in ONE contiguous buffer, stepped by ``rnh_adam_step``: one launch per run of parameters that received
  a gradient (two runs for RefineNet: ``refine_block.prelu.weight`` never gets one, quirk Q1) instead of
  ``torch.optim.Adam``'s multi-tensor chains.  ``state_dict()`` / ``load_state_dict()`` keep ``torch.optim.Adam``'s
  format, so the reference's checkpoints (``base_trainer.py:229-245``) go in and out unchanged.
* ``psnr_ssim``: denormalize + PSNR + SSIM of all image pairs of a step through ``rnh_metrics_psnr_ssim``.

No fallback: tensors that are not fp32 on a HIP device raise.
"""
import ctypes as C
import math

import torch

from . import lib as L


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _need_hip(t, what):
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32):
        raise L.HipKernelError(f'{what}: needs fp32 tensors on a HIP device (there is no CPU path), got '
                               f'{getattr(t, "dtype", type(t))} on {getattr(t, "device", "?")}')


# ------------------------------------------------------------------------------------------------------------
# metrics
# ------------------------------------------------------------------------------------------------------------
def ssim_window_1d():
    """The 11 weights w with outer(w, w) == the reference's 2-D window (src/model/metrics.py:67-84: per-axis factor
    exp(-((x - 5) / (2 * 1.5)) ** 2), the product normalised to sum 1, so each axis is normalised to sum 1)."""
    g = [math.exp(-((x - 5) / (2 * 1.5)) ** 2) for x in range(11)]
    s = sum(g)
    return (C.c_float * 11)(*[v / s for v in g])


_WINDOW = ssim_window_1d()
_ws_cache = {}


def psnr_ssim(out, tgt, P, cps, H, W, denorm=None, max_value=255.0, value_range=255.0, want_ssim=True):
    """out, tgt: fp32 device tensors holding P contiguous H x W planes each (cps planes per sample).  ``denorm`` =
    (mean, std) applies the reference's denormalize to both first.  Returns the result vector of
    rnh_metrics_psnr_ssim: [mean PSNR, mean SSIM, PSNR per sample (P/cps), SSIM per plane (P), MSE per plane (P)]."""
    _need_hip(out, 'psnr_ssim')
    _need_hip(tgt, 'psnr_ssim')
    if not (out.is_contiguous() and tgt.is_contiguous()) or out.numel() != P * H * W or tgt.numel() != P * H * W:
        raise L.HipKernelError(f'psnr_ssim: expected two contiguous tensors of {P}x{H}x{W} elements, got {tuple(out.shape)} / {tuple(tgt.shape)}')
    lib = L.load()
    key = (out.device, P, H, W)
    ws = _ws_cache.get(key)
    if ws is None:
        _ws_cache.clear()
        ws = _ws_cache[key] = torch.empty(lib.rnh_metrics_ws_floats(P, H, W), dtype=torch.float32, device=out.device)
    res = torch.empty(2 + P // max(cps, 1) + 2 * P, dtype=torch.float32, device=out.device)
    mean, std = denorm if denorm is not None else (0.0, 1.0)
    L.check(lib.rnh_metrics_psnr_ssim(C.c_void_p(out.data_ptr()), C.c_void_p(tgt.data_ptr()), P, cps, H, W, int(denorm is not None),
                                      int(bool(want_ssim)), mean, std, max_value, value_range, _WINDOW, C.c_void_p(ws.data_ptr()),
                                      C.c_void_p(res.data_ptr()), _stream(out.device)), 'rnh_metrics_psnr_ssim')
    return res


# ------------------------------------------------------------------------------------------------------------
# Adam
# ------------------------------------------------------------------------------------------------------------
class FlatAdam(torch.optim.Optimizer):
    """Adam on flat HIP buffers; constructor arguments, ``param_groups`` and ``state`` as ``torch.optim.Adam``."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, *, maximize=False,
                 foreach=None, capturable=False, differentiable=False, fused=None):
        if not 0.0 <= lr:
            raise ValueError(f'Invalid learning rate: {lr}')
        if not 0.0 <= eps:
            raise ValueError(f'Invalid epsilon value: {eps}')
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError(f'Invalid beta parameter at index 0: {betas[0]}')
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError(f'Invalid beta parameter at index 1: {betas[1]}')
        if not 0.0 <= weight_decay:
            raise ValueError(f'Invalid weight_decay value: {weight_decay}')
        if amsgrad or maximize or capturable or differentiable:
            raise ValueError('FlatAdam implements plain Adam (amsgrad / maximize / capturable / differentiable are not supported)')
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False, foreach=foreach,
                        capturable=False, differentiable=False, fused=fused, decoupled_weight_decay=False)
        super().__init__(params, defaults)
        self._flat = {}       # group index -> dict(p, m, v, offsets)
        self.launches = 0     # kernel launches of the last step() (for tests / the bench report)

    # -- flat storage -------------------------------------------------------------------------------------------
    def _adopt(self, gi, group):
        """Make every parameter of the group (and its moments, once they exist) a view of the group's flat buffers.
        Re-done whenever a tensor was replaced behind our back (module.to(), load_state_dict)."""
        params = group['params']
        fl = self._flat.get(gi)
        dev = params[0].device
        for p in params:
            _need_hip(p, 'FlatAdam')
            if p.device != dev:
                raise L.HipKernelError('FlatAdam: all parameters of a group must live on one device')
        if fl is None or fl['p'].device != dev or fl['n'] != sum(p.numel() for p in params):
            n = sum(p.numel() for p in params)
            fl = self._flat[gi] = dict(n=n, p=torch.empty(n, dtype=torch.float32, device=dev),
                                       m=torch.zeros(n, dtype=torch.float32, device=dev),
                                       v=torch.zeros(n, dtype=torch.float32, device=dev), g=None)
        off = 0
        for p in params:
            k = p.numel()
            if p.data.data_ptr() != fl['p'].data_ptr() + 4 * off or not p.data.is_contiguous():
                view = fl['p'][off:off + k].view(p.shape)
                view.copy_(p.data)
                p.data = view
            st = self.state.get(p)
            if st:
                for name, buf in (('exp_avg', fl['m']), ('exp_avg_sq', fl['v'])):
                    if st[name].data_ptr() != buf.data_ptr() + 4 * off:
                        view = buf[off:off + k].view(p.shape)
                        view.copy_(st[name])
                        st[name] = view
            off += k
        return fl

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.load()
        self.launches = 0
        for gi, group in enumerate(self.param_groups):
            if not group['params']:
                continue
            fl = self._adopt(gi, group)
            b1, b2 = group['betas']
            runs, off = [], 0          # [offset, count, grad pointer, step]
            for p in group['params']:
                k = p.numel()
                g = p.grad
                if g is not None:
                    if g.is_sparse:
                        raise RuntimeError('Adam does not support sparse gradients')
                    _need_hip(g, 'FlatAdam (gradient)')
                    st = self.state[p]
                    if not st:
                        st['step'] = torch.tensor(0.0, dtype=torch.float32)
                        st['exp_avg'] = fl['m'][off:off + k].view(p.shape)
                        st['exp_avg_sq'] = fl['v'][off:off + k].view(p.shape)
                    st['step'] += 1
                    t = int(st['step'])
                    gp = g.data_ptr() if g.is_contiguous() else None
                    if gp is None or (gp - (fl['p'].data_ptr() + 4 * off)) % 16:
                        # a gradient that is not laid out like the parameters: stage it at the parameter's offset
                        if fl['g'] is None:
                            fl['g'] = torch.empty_like(fl['p'])
                        fl['g'][off:off + k].view(p.shape).copy_(g)
                        gp = fl['g'].data_ptr() + 4 * off
                    last = runs[-1] if runs else None
                    if last is not None and last[0] + last[1] == off and last[2] + 4 * last[1] == gp and last[3] == t:
                        last[1] += k
                    else:
                        runs.append([off, k, gp, t])
                off += k
            for o, k, gp, t in runs:
                L.check(lib.rnh_adam_step(C.c_void_p(fl['p'].data_ptr() + 4 * o), C.c_void_p(gp), C.c_void_p(fl['m'].data_ptr() + 4 * o),
                                          C.c_void_p(fl['v'].data_ptr() + 4 * o), k, t, group['lr'], b1, b2, group['eps'],
                                          group['weight_decay'], _stream(fl['p'].device)), 'rnh_adam_step')
                self.launches += 1
        return loss
