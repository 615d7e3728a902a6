"""Are `nt` gate stores of the F(4x4) cell kernel (-DW4_GATES_NT) safe?  The same cell launch at config 2's size through the product build and the nt
build, alternating, `reps` times each: gates / c' / h' of every launch bit-compared with the first product launch (the two builds differ in the
cache policy of four stores only).   python tools/experiments/r05_wino44_nt_soak.py [reps]   (needs hipvsr/librefinenet_nt.so)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd')
sys.path.insert(0, ROOT); sys.path.insert(0, PKG)
import torch
from hipvsr import lib as L
from hipvsr.hip_ops import HipOps
from hipvsr.plans import NetPlans, Src
from hipvsr.spec import state_dict_spec
from oracle import refinenet_oracle as orc

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B, H, W = 8, 128, 128
dev = torch.device('cuda:0')
cfg = orc.exp1_x4_config()
P, ops = NetPlans(cfg), HipOps(dev)
spec = state_dict_spec(cfg)
plan = P.lstm[('forward', 1)]['full']
g = torch.Generator('cpu').manual_seed(3)
ops.pack(plan, (torch.randn(*spec[plan.wkey], generator=g) * 0.03).to(dev), (torch.randn(*spec[plan.bkey], generator=g) * 0.1).to(dev))
x, h, c = (torch.randn(B, H, W, 64, generator=g).to(dev) for _ in range(3))
vx, vh = ops.wino44_v(B, H, W, 64)[0], ops.wino44_v(B, H, W, 64)[0]
ops.wino44_transform(Src(x), B, H, W, vx)
ops.wino44_transform(Src(h), B, H, W, vh)
nt = ctypes.CDLL(os.path.join(PKG, 'hipvsr', 'librefinenet_nt.so'))
nt.rnh_wino44_cell.argtypes = ops.lib.rnh_wino44_cell.argtypes


def launch(lib):
    ho, co = (torch.full((B, H, W, 64), float('nan'), device=dev) for _ in range(2))
    go = torch.full((B, H, W, 256), float('nan'), device=dev)
    real, ops.lib = ops.lib, lib
    try:
        ops.wino44_cell(plan, [vx, vh], B, H, W, dict(hd=64, c_prev=c, h_out=ho, c_out=co, gates_out=go))
    finally:
        ops.lib = real
    return go, co, ho


class Both:                                     # the nt build's cell entry, everything else from the product library
    def __init__(self, a, b):
        self.a, self.b = a, b

    def __getattr__(self, k):
        return getattr(self.b if k == 'rnh_wino44_cell' else self.a, k)


ref = launch(ops.lib)
torch.cuda.synchronize()
bad = {'product': 0, 'nt': 0}
for i in range(reps):
    for name, lib in (('product', ops.lib), ('nt', Both(ops.lib, nt))):
        out = launch(lib)
        torch.cuda.synchronize()
        bad[name] += int(not all(torch.equal(a, b) for a, b in zip(out, ref)))
print(f'{reps} launches each at N={B} {H}x{W}: launches that differ from the first product launch in any bit of gates / c / h: {bad}')
