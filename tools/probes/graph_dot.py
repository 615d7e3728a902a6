"""Probe: capture the training step WITH the helper-stream branch (RNH_ASIDE_CAPTURE=1), dump the HIP graph as DOT (hipGraphDebugDotPrint) and replay it
against the eager step several times.  usage: graph_dot.py <out.dot> [f32|bf16]"""
import os, sys
os.environ['RNH_ASIDE_CAPTURE'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd')):
    sys.path.insert(0, p)
import torch
from oracle import refinenet_oracle as orc
from hipvsr.step_tail import FlatAdam
from hipvsr import graph as G
from src.model.nets import RefineNet
from src.runner.trainers import AcdcVSRRefineNetTrainer
out, dtype = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'f32')
dev = torch.device('cuda:0')
cfg = orc.Config(in_channels=1, out_channels=1, num_features=[16, 16], num_stages=2, refine_window_size=5, upscale_factor=4,
                 update_memory=True, num_updated_frames=2, positional_encoding=True)
sd = orc.init_state_dict(cfg, seed=8)
g = torch.Generator('cpu').manual_seed(5)
n = 4
batches = [([torch.randn(n, 1, 16, 16, generator=g).to(dev) for _ in range(7)], [torch.randn(n, 1, 64, 64, generator=g).to(dev) for _ in range(3)],
            (torch.rand(n, 7, 1, generator=g) * 2 - 1).to(dev)) for _ in range(6)]
Orig = torch.cuda.CUDAGraph
graphs = []
def Dbg(*a, **k):
    g_ = Orig(keep_graph=True)
    graphs.append(g_)
    return g_
torch.cuda.CUDAGraph = Dbg
runs = {}
for graph in (False, True):
    net = RefineNet(**cfg)
    net.load_state_dict(sd)
    net = net.to(dev).set_compute_dtype(dtype).train()
    tr = object.__new__(AcdcVSRRefineNetTrainer)
    tr.net, tr.loss_fns, tr.metric_fns, tr.graph, tr._graphed = net, [torch.nn.L1Loss()], [], graph, None
    tr.loss_weights = torch.tensor([1.0], device=dev)
    tr.optimizer = FlatAdam(net.parameters(), lr=1e-3)
    hist = []
    for xs, ys, pc in batches:
        outs, loss, _ = tr.train_step(xs, ys, pc)
        torch.cuda.synchronize()
        hist.append({k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
        hist[-1].update({'PARAM ' + k: p.detach().clone() for k, p in net.named_parameters()})
        ent = net._engine().ops._ws.get(('halo', 'dR1p'))
        if ent is not None:
            (shape, dt_, lo, hi), t = ent
            print('   graph' if graph else '   eager', 'step', len(hist) - 1, 'halo of dR1p nonzero entries:', int((t[:lo] != 0).sum()) + int((t[hi:] != 0).sum()),
                  'NaN in the middle:', int(torch.isnan(t[lo:hi]).sum()))
    runs[graph] = hist
if graphs:
    import ctypes
    hip = ctypes.CDLL('libamdhip64.so')
    h = graphs[0].raw_cuda_graph()
    hip.hipGraphDebugDotPrint.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint]
    rc = hip.hipGraphDebugDotPrint(ctypes.c_void_p(h), os.path.abspath(out).encode(), 1 << 0)      # hipGraphDebugDotFlagsVerbose
    print('hipGraphDebugDotPrint rc', rc, out, os.path.getsize(out) if os.path.exists(out) else 'MISSING')
first = True
for i, (ga, gb) in enumerate(zip(runs[False], runs[True])):
    bad = [(k, float((ga[k] - gb[k]).abs().max())) for k in ga if not torch.equal(ga[k], gb[k])]
    print('step', i, 'differing:', [b[0].split('.')[0] + '..' + b[0].split('.')[-1] + f':{b[1]:.1e}' for b in bad][:8])
    if bad and first:
        first = False
        for k, d in bad:
            ne = (ga[k] != gb[k])
            idx = ne.flatten().nonzero().flatten()
            print(f'    {k}: {int(ne.sum())} of {ne.numel()} elements differ, max {d:.2e} of max|g| {float(ga[k].abs().max()):.2e}; first flat indices {idx[:6].tolist()} last {idx[-3:].tolist()} shape {tuple(ga[k].shape)}')
