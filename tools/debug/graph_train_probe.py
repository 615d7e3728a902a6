"""Which part of the training step breaks HIP graph capture?  python tools/debug/graph_train_probe.py <variant>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd')
sys.path[:0] = [ROOT, PKG]
import torch
from oracle import refinenet_oracle as orc
from src.model.nets import RefineNet
from src.runner.trainers import AcdcVSRRefineNetTrainer
variant = sys.argv[1]
dev = torch.device('cuda:0')
cfg = orc.Config(in_channels=1, out_channels=1, num_features=[16, 16], num_stages=2, refine_window_size=5, upscale_factor=4,
                 update_memory=True, num_updated_frames=2, positional_encoding=True)
net = RefineNet(**cfg).to(dev).train()
if 'one' in variant:
    os.environ['RNH_LSTM_STREAMS'] = 'one'
tr = object.__new__(AcdcVSRRefineNetTrainer)
tr.net, tr.loss_fns, tr.metric_fns = net, [torch.nn.L1Loss()], []
tr.loss_weights = torch.tensor([1.0], device=dev)
g = torch.Generator('cpu').manual_seed(5)
n = 2
xs = list(torch.stack([torch.randn(n, 1, 16, 16, generator=g) for _ in range(7)]).to(dev).unbind(0))
ys = list(torch.stack([torch.randn(n, 1, 64, 64, generator=g) for _ in range(3)]).to(dev).unbind(0))
pc = (torch.rand(n, 7, 1, generator=g) * 2 - 1).to(dev)
params = list(net.parameters())

_keep = []
def body():
    if variant.startswith('v'):
        eng = net._engine()
        pd = {k: p.detach() for k, p in net.named_parameters()}
        with torch.no_grad():
            O, ctx = eng.forward(pd, xs, pc, need_grad=True)
            if variant == 'v1':
                _keep.append(ctx)
                return O.sum()
            if variant == 'v2':
                del ctx
                return O.sum()
            if variant == 'v3':
                _keep.append(ctx)
                dO = torch.full_like(O, 1e-3)
                flat = torch.zeros(sum(p.numel() for p in params), device=dev)
                return O.sum() + flat.sum() + dO.sum()
            if variant == 'v4':          # free the saved activations stage by stage like the backward does
                for i in range(len(ctx.stages)):
                    ctx.stages[i] = None
                return O.sum()
    if 'engine' in variant:          # the engine's forward + hand-written backward without autograd (main thread only)
        eng = net._engine()
        pd = {k: p.detach() for k, p in net.named_parameters()}
        with torch.no_grad():
            O, ctx = eng.forward(pd, xs, pc, need_grad=True)
            stop = os.environ.get('PROBE_STOP')
            dO = torch.full_like(O, 1e-3)
            flat = torch.zeros(sum(p.numel() for p in params), device=dev)
            eng.backward(pd, ctx, dO, flat=flat)
        return flat.sum()
    outs = net(xs, pc)
    if 'fwdonly' in variant:
        return outs
    if 'simpleloss' in variant:
        loss = outs.packed.abs().mean()
    else:
        losses = tr._compute_losses(outs, ys)
        loss = (torch.stack(losses) * tr.loss_weights).sum()
    if 'nobwd' not in variant:
        loss.backward()
    return loss

# PROBE_N=k: the backward is ABORTED (exception) before its (k+1)-th kernel-wrapper call - nothing runs on dummy operands
_limit = int(os.environ.get('PROBE_N', '-1'))
_state = dict(n=0, on=False, log=[])
class _Stop(Exception):
    pass
if 'engine' in variant:
    ops = net._engine().ops
    for name in ('conv', 'wgrad', 'add', 'lstm_gates_bwd', 'inconv_bwd', 'uptail_compose', 'uptail_xcorr', 'uptail_expand', 'uptail_wcontract',
                 'uptail_dgrad', 'refine_xcol_wgrad', 'zeros', 'pack', 'fork', 'join'):
        orig = getattr(ops, name)
        def wrap(*a, _orig=orig, _name=name, **k):
            if _state['on']:
                if 0 <= _limit <= _state['n']:
                    raise _Stop()
                _state['n'] += 1
                _state['log'].append(_name)
            return _orig(*a, **k)
        setattr(ops, name, wrap)
    eng0 = net._engine()
    _bwd = eng0.backward
    def bwd(*a, **k):
        _state['on'] = True
        _state['n'] = 0
        _state['log'] = []
        try:
            return _bwd(*a, **k)
        except _Stop:
            _state['on'] = False
            forks = _state['log'].count('fork') - _state['log'].count('join')
            if forks > 0:
                ops.join(4)    # rejoin the side streams if the abort came between fork and join
            return None
        finally:
            _state['on'] = False
    eng0.backward = bwd

st = torch.cuda.Stream(dev)
st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    for _ in range(2):
        for p in params:
            p.grad = None
        body()
st.synchronize()
for p in params:
    p.grad = None
print('warm-up done', _state['n'], 'backward ops:', ' '.join(_state['log'][:80]), flush=True)
gr = torch.cuda.CUDAGraph()
mode = 'relaxed' if 'relaxed' in variant else ('thread_local' if 'tl' in variant else 'global')
with torch.cuda.graph(gr, stream=st, capture_error_mode=mode):
    out = body()
print('captured', variant, flush=True)
gr.replay()
torch.cuda.synchronize()
print('replayed', variant, float(out.detach()) if torch.is_tensor(out) else 'ok', flush=True)
