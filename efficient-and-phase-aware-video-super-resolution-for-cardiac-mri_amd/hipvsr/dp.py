"""Data-parallel gradient exchange: one all-reduce per step over the flat gradient buffer.

The reference has no distributed code at all; this is the multi-GPU row of the hot path:
one process per GPU, samples sharded over ranks, weights replicated, and the 2.89 M-parameter (11.6 MB fp32)
gradient averaged with a single collective (RCCL over xGMI when the backend is "nccl"; "gloo" in CPU tests).
The engine's backward writes every parameter gradient as a view of one contiguous buffer, so no flatten copy
is needed; if the views were replaced (e.g. by gradient accumulation hooks) the buffer is rebuilt.
"""
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def allreduce_log(log, count, device):
    """Sum the running log sums and the sample count of an epoch over all ranks (one small collective per epoch), so
    that ReduceLROnPlateau, Monitor.is_best and the early stop see the same numbers everywhere - a rank that stopped
    early on its own shard's log would leave the others waiting in the next gradient all-reduce."""
    if world() == 1:
        return log, count
    keys = sorted(log)
    t = torch.tensor([float(log[k]) for k in keys] + [float(count)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    vals = t.tolist()
    return {k: v for k, v in zip(keys, vals[:-1])}, vals[-1]


def broadcast_parameters(module, src=0):
    """Make every rank start from rank ``src``'s weights."""
    if world() == 1:
        return
    with torch.no_grad():
        flat = torch.cat([p.detach().reshape(-1) for p in module.parameters()])
        dist.broadcast(flat, src)
        off = 0
        for p in module.parameters():
            p.copy_(flat[off:off + p.numel()].view_as(p))
            off += p.numel()


_AR_EVENTS = []           # (start, end) HIP events around the gradient all-reduce of the most recent steps (GPU tensors only; at most 64 pairs)


def allreduce_ms():
    """Mean HIP-event time of the recorded gradient all-reduces (ms), or None if none ran; clears the record.  Synchronises the device."""
    if not _AR_EVENTS:
        return None
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in _AR_EVENTS]
    _AR_EVENTS.clear()
    return sum(ms) / len(ms)


def allreduce_gradients(module, average=True, force=False):
    """Sum (and average) the gradients of ``module`` over all ranks with ONE collective.  Returns the number
    of bytes reduced (0 when world size is 1, unless ``force`` runs the collective anyway - used to test the
    RCCL path on a single GPU)."""
    ws = world()
    if ws == 1 and not (force and dist.is_initialized()):
        return 0
    params = list(module.parameters())
    flat = getattr(module, '_flat_grad', None)
    in_place = flat is not None
    if in_place:
        off = 0
        for p in params:
            if p.grad is not None and p.grad.data_ptr() != flat.data_ptr() + 4 * off:
                in_place = False
                break
            off += p.numel()
        in_place = in_place and off == flat.numel()
    if not in_place:
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    timed = flat.is_cuda and len(_AR_EVENTS) < 64
    if timed:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat.mul_(1.0 / ws)
    if timed:
        ev[1].record()
        _AR_EVENTS.append(ev)
    if not in_place:
        off = 0
        for p in params:
            if p.grad is not None:
                p.grad.copy_(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
    return flat.numel() * 4
