"""Bridge between torch.autograd and the engine: one Function for the whole network (forward = the engine's
forward, backward = the engine's hand-written backward) and one for the fused loss + gradient kernel."""
import torch

from . import lib as L
from .hip_ops import packed_view


class HipOutputs(tuple):
    """Return value of RefineNet.forward: the reference's tuple[3*S] of list[T] of (N, C, sH, sW) tensors
    (reference src/model/nets/refine_net.py:135), plus ``packed``: the single (S, 3, T*N, sH, sW, C) tensor all
    of them are views of, which the fused loss kernel consumes in one launch."""
    packed = None
    targets_key = None


class RefineNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, inputs, pos_codes, *params):
        eng = module._engine()
        names = module._param_names
        pd = {n: p.detach() for n, p in zip(names, params)}
        need = any(ctx.needs_input_grad[3:])          # False under torch.no_grad() and for frozen parameters
        O_all, ectx = eng.forward(pd, inputs, pos_codes, need_grad=need,
                                  last_only=bool(getattr(module, 'last_group_only', False)) and not need)
        ctx.module, ctx.ectx, ctx.pd = module, ectx, pd
        return O_all

    @staticmethod
    def backward(ctx, dO_all):
        module = ctx.module
        eng = module._engine()
        if ctx.ectx is None:
            raise RuntimeError('RefineNet backward called without a saved context')
        names = module._param_names
        n = sum(ctx.pd[k].numel() for k in names)
        flat = torch.zeros(n, dtype=torch.float32, device=dO_all.device)
        grads = eng.backward(ctx.pd, ctx.ectx, dO_all.contiguous(), flat=flat)
        ctx.ectx = None
        module._flat_grad = flat
        return (None, None, None) + tuple(grads[k] for k in names)


class FusedLossFn(torch.autograd.Function):
    """loss[g*T+i] = mean(l(O[g,i] - Y[i])) for all groups in one launch; backward recomputes l' scaled by the
    incoming gradient in one more launch (rnh_loss_fwd_bwd)."""

    @staticmethod
    def forward(ctx, ops, o, y, G, T, kind, eps):
        loss, _ = ops.loss(o, y, G, T, kind, eps, None, want_grad=False)
        ctx.ops, ctx.meta = ops, (G, T, kind, eps)
        ctx.save_for_backward(o, y)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        o, y = ctx.saved_tensors
        G, T, kind, eps = ctx.meta
        _, d_o = ctx.ops.loss(o, y, G, T, kind, eps, gloss.contiguous().float(), want_grad=True)
        return None, d_o, None, None, None, None, None


class LossTotalFn(torch.autograd.Function):
    """total = sum_g w[g] * mean_i loss[g*T + i]: the trainer's discounted deep-supervision sum in one launch (and one more for
    its gradient) instead of the ~30 element-wise / reduction kernels per direction that the list arithmetic costs."""

    @staticmethod
    def forward(ctx, ops, per_pair, w, G, T):
        ctx.ops, ctx.meta = ops, (G, T)
        ctx.save_for_backward(w)
        return ops.loss_total(per_pair.contiguous(), w, G, T).reshape(())

    @staticmethod
    def backward(ctx, gtot):
        (w,) = ctx.saved_tensors
        G, T = ctx.meta
        return None, ctx.ops.loss_total(gtot.reshape(1).contiguous().float(), w, G, T, backward=True), None, None, None


def discounted_total(outputs, per_pair, discounts, T):
    """The training loss of one loss function from the fused per-pair values (None if they did not come from the HIP path)."""
    ops = getattr(outputs, 'ops', None)
    if ops is None or per_pair is None or not per_pair.is_cuda:
        return None
    key = ('loss_discounts', tuple(discounts))
    cache = ops.__dict__.setdefault('_const_cache', {})
    if key not in cache:
        cache[key] = torch.tensor(list(discounts), dtype=torch.float32, device=per_pair.device)
    return LossTotalFn.apply(ops, per_pair, cache[key], len(discounts), T)


def fused_losses(outputs, targets, loss_fn, last_only=False, per_sample=False):
    """Per-(group, frame) loss values [G*T] through the fused kernel, or None if this combination is not
    served by it (then the caller falls back to calling loss_fn per pair, like the reference does).
    last_only: the T values of the last group alone (what the predictor and the evaluation branch consume);
    with per_sample the (T, N) values of every sample on its own (a predictor that runs N cines at once)."""
    packed = getattr(outputs, 'packed', None)
    if packed is None or not packed.is_cuda:
        return None
    name = type(loss_fn).__name__
    if name == 'L1Loss' and getattr(loss_fn, 'reduction', 'mean') == 'mean':
        kind, eps = L.LOSS_L1, 0.0
    elif name == 'CharbonnierLoss':
        kind, eps = L.LOSS_CHARBONNIER, float(loss_fn.epsilon)
    else:
        return None
    S, three, TN = packed.shape[:3]
    T = len(targets)
    y = packed_view(list(targets))                                      # (T, N, C, sH, sW)
    if y.shape[2] != 1:
        y = y.permute(0, 1, 3, 4, 2)
    y = y.contiguous().float()
    ops = outputs.ops
    if last_only and per_sample:
        N = TN // T
        return FusedLossFn.apply(ops, packed[S - 1, three - 1].reshape(TN, -1), y.reshape(TN, -1), 1, TN, kind, eps).view(T, N)
    if last_only:
        return FusedLossFn.apply(ops, packed[S - 1, three - 1].reshape(T, -1), y.reshape(T, -1), 1, T, kind, eps)
    return FusedLossFn.apply(ops, packed.reshape(S * three * T, -1), y.reshape(T, -1), S * three, T, kind, eps)
