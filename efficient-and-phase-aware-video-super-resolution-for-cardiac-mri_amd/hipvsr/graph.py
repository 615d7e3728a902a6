"""Whole-cycle inference replayed from a HIP graph.

The reference's predictor (src/runner/predictors/acdc_vsr_refinenet_predictor.py:31-109) runs ``net(inputs, pos_codes)``
under ``torch.no_grad()`` on ONE whole cardiac cycle at a time (batch 1, F = T_cycle + 2U frames of e.g. 54 x 64 pixels):
about a thousand sub-20-microsecond launches, i.e. bound by the host's launch rate, not by the GPU.  ``GraphedForward``
captures the engine's forward once per input shape - the layer/frame wavefront of the ConvLSTM on its 2L side streams
becomes parallel branches of the graph - and replays it with one ``hipGraphLaunch``.

Static buffers: the inputs are copied into the graph's input buffer (one small D2D copy), the outputs are views of the
graph's output buffer and are overwritten by the next call with the same shape; weights are read (and re-packed) inside
the graph from the parameters' storage, so ``load_state_dict`` / in-place updates are seen by later replays.

Every address a graph has baked in stays valid for the graph's life: tensors allocated during the capture live in the
graph's private pool; the engine's persistent buffers that exist before the capture (packed weight slabs, index maps,
per-stream scratch: created by the eager warm-up run) are owned by the net's ``HipOps`` and never freed - a scratch buffer
that a later, larger call outgrows is retired instead (``HipOps._workspace``, ``graph_captures``).  Replays, eager
forwards and training steps of the same net may therefore interleave freely (tests/test_predictor.py).
"""
import torch

from .hip_ops import packed_view


class _Entry:
    __slots__ = ('graph', 'x', 'pos', 'outputs', 'replays')


class GraphedForward:
    def __init__(self, net, max_graphs=8):
        self.net, self.max_graphs = net, max_graphs
        self._entries = {}
        self._stream = None
        self._versions = None

    def _param_key(self):
        # a parameter that was REPLACED (not updated in place) invalidates the captured pointers
        return tuple(p.data_ptr() for p in self.net.parameters())

    def _capture(self, inputs, pos_codes):
        dev = inputs[0].device
        if dev.type != 'cuda':
            raise RuntimeError(f'GraphedForward needs inputs on a HIP device, got {dev}')
        if self._stream is None:
            self._stream = torch.cuda.Stream(dev)
        e = _Entry()
        e.x = torch.stack([x.detach().to(dev, torch.float32) for x in inputs], dim=0).contiguous()      # (F, N, Cin, H, W)
        e.pos = pos_codes.detach().to(dev, torch.float32).clone()
        static_in = [e.x[k] for k in range(e.x.shape[0])]
        st = self._stream
        self.net._engine().ops.graph_captures += 1      # from here on the engine retires scratch buffers instead of freeing them
        st.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(st), torch.no_grad():
            self.net(static_in, e.pos)             # eager run on the capture stream: plans, index maps, workspaces exist afterwards
        st.synchronize()
        e.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(e.graph, stream=st), torch.no_grad():
            e.outputs = self.net(static_in, e.pos)
        e.replays = 0
        return e

    @torch.no_grad()
    def __call__(self, inputs, pos_codes):
        net = self.net
        if net.training:
            raise RuntimeError('GraphedForward serves evaluation (net.eval()); the training step is not captured')
        pk = self._param_key()
        if pk != self._versions:
            self._entries.clear()
            self._versions = pk
        key = (len(inputs), tuple(inputs[0].shape), tuple(pos_codes.shape), bool(getattr(net, 'last_group_only', False)))
        e = self._entries.get(key)
        if e is None:
            if len(self._entries) >= self.max_graphs:
                self._entries.pop(next(iter(self._entries)))
            e = self._entries[key] = self._capture(inputs, pos_codes)
        x = packed_view([t.detach() for t in inputs], e.x.device)
        e.x.copy_(x.view_as(e.x))
        e.pos.copy_(pos_codes)
        e.graph.replay()
        e.replays += 1
        return e.outputs


class _StepEntry:
    __slots__ = ('graph', 'x', 'y', 'pos', 'outputs', 'loss', 'losses', 'grads', 'flat', 'replays')


class GraphedTrainStep:
    """forward + loss + backward of a training step replayed from a HIP graph (SURVEY.md section 8a row A6 "(later)
    HIP-graph"; reference src/runner/trainers/acdc_vsr_refinenet_trainer.py:41-46: ``net(inputs, pos_codes)``,
    ``_compute_losses``, ``loss.backward()``).

    Measured (profiles/ARCHIVE/r02_e_train_shape.txt, r02_l_train_shape.txt): at the reference's own training shape
    (configs/train/refine_net/exp1_x4.yaml:21-33: batch 16, 32 x 32 crops) the step is GPU-bound, not launch-bound - fp32 50.9 ms
    eager against 51.3 ms replayed, bf16 24.2 against 27.0 (slower) - so the trainer does NOT use this class unless asked to
    (trainer kwarg ``graph: true``); it is kept for smaller batches, where a step is ~1 500 launches of a few microseconds each.
    The whole step up to the gradients is captured once per input shape - the ConvLSTM wavefront's side streams become parallel
    branches - and replayed with one launch; what stays outside: the copy of the batch into the graph's static input buffers,
    the gradient all-reduce (one collective; single rank only, the trainer refuses the combination), the optimizer step (two
    launches of the flat Adam whose step count and learning rate are host-side arguments) and metrics / logging.

    Gradients: the capture runs with ``p.grad is None``, so autograd adopts the engine's gradient tensors - views of one
    flat buffer in the graph's pool - and a replay overwrites them in place; there is no ``zero_grad`` between replays and
    nothing accumulates.  Replay and eager step launch the same kernels with the same arguments (both take the fused
    loss-total launch, hipvsr.autograd.LossTotalFn), which is why tests/test_predictor.py can ask for bit-identical results;
    a loss function outside the fused path would make the two differ in summation order."""

    def __init__(self, trainer, max_graphs=4):
        self.tr, self.max_graphs = trainer, max_graphs
        self._entries = {}
        self._stream = None
        self._versions = None

    def _body(self, e):
        tr = self.tr
        outputs = tr.net(e.x, e.pos)
        losses = tr._compute_losses(outputs, e.y)
        loss = tr._total_loss(losses) if hasattr(tr, '_total_loss') else (torch.stack(losses) * tr.loss_weights).sum()
        if hasattr(tr, '_backward'):
            tr._backward(loss)                     # (the seed gradient is allocated in the eager warm-up, outside the capture)
        else:
            loss.backward()
        return outputs, loss, losses

    def _capture(self, inputs, targets, pos_codes):
        net = self.tr.net
        dev = inputs[0].device
        if dev.type != 'cuda':
            raise RuntimeError(f'GraphedTrainStep needs inputs on a HIP device, got {dev}')
        if self._stream is None:
            self._stream = torch.cuda.Stream(dev)
        e = _StepEntry()
        xb = torch.stack([x.detach().to(dev, torch.float32) for x in inputs], dim=0).contiguous()
        yb = torch.stack([y.detach().to(dev, torch.float32) for y in targets], dim=0).contiguous()
        e.x, e.y = list(xb.unbind(0)), list(yb.unbind(0))
        e.pos = pos_codes.detach().to(dev, torch.float32).clone()
        params = list(net.parameters())
        st = self._stream
        net._engine().ops.graph_captures += 1      # scratch buffers are retired, not freed, from here on (HipOps._workspace)
        st.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(st):
            for _ in range(2):                     # eager warm-up on the capture stream: plans, index maps, workspaces, autograd
                for p in params:
                    p.grad = None
                self._body(e)
        st.synchronize()
        for p in params:
            p.grad = None
        e.graph = torch.cuda.CUDAGraph()
        # thread-local error mode: under torch.distributed RCCL's watchdog thread makes HIP calls of its own (event queries), which a capture in the
        # default global mode would reject as 'operation not permitted when stream is capturing'
        import torch.distributed as dist
        mode = 'thread_local' if dist.is_available() and dist.is_initialized() else 'global'
        with torch.cuda.graph(e.graph, stream=st, capture_error_mode=mode):
            e.outputs, e.loss, e.losses = self._body(e)
        e.grads = [p.grad for p in params]
        e.flat = net._flat_grad
        e.replays = 0
        return e

    def __call__(self, inputs, targets, pos_codes):
        net = self.tr.net
        if not net.training:
            raise RuntimeError('GraphedTrainStep serves training (net.train())')
        pk = tuple(p.data_ptr() for p in net.parameters())
        if pk != self._versions:                   # a parameter was REPLACED (load_state_dict copies in place and keeps them)
            self._entries.clear()
            self._versions = pk
        key = (len(inputs), tuple(inputs[0].shape), len(targets), tuple(targets[0].shape), tuple(pos_codes.shape),
               getattr(net, 'compute_dtype', 'f32'))
        e = self._entries.get(key)
        if e is None:
            if len(self._entries) >= self.max_graphs:
                self._entries.pop(next(iter(self._entries)))
            e = self._entries[key] = self._capture(inputs, targets, pos_codes)
        xv, yv = packed_view([t.detach() for t in inputs], e.pos.device), packed_view([t.detach() for t in targets], e.pos.device)
        packed_view(e.x).copy_(xv)
        packed_view(e.y).copy_(yv)
        e.pos.copy_(pos_codes)
        e.graph.replay()
        e.replays += 1
        for p, g in zip(net.parameters(), e.grads):   # this shape's gradient tensors (another shape may have run in between)
            p.grad = g
        net._flat_grad = e.flat
        return e.outputs, e.loss, e.losses
