import os, sys
sys.argv = ['x', 'zzz']       # filter matching nothing: only set-up
__file__ = os.path.abspath('tools/kbench_bf16.py')
exec(open('tools/kbench_bf16.py').read().split("px = N * H * W")[0])
flt = ''
px = N * H * W
pl = P.lstm[('forward', 1)]
x, hp, cp = R(N, H, W, 64), R(N, H, W, 64), R(N, H, W, 64, dtype=torch.float32)
ho, co, go = ops.empty(N, H, W, 64, dtype=bf), ops.empty(N, H, W, 64), ops.empty(N, H, W, 256, dtype=bf)
f1 = lambda: ops.conv(pl['full'], [Src(x), Src(hp)], N, H, W, lstm=dict(hd=64, c_prev=cp, h_out=ho, c_out=co, gates_out=go))
f0 = lambda: ops.conv(pl['full'], [Src(x), Src(hp)], N, H, W, lstm=dict(hd=64, c_prev=cp, h_out=ho, c_out=co, gates_out=None))
for rep in range(3):
    timeit('nogates', f0, 2.0 * px * 256 * 1152, px * 896, 50)
    timeit('gates', f1, 2.0 * px * 256 * 1152, px * 1408, 50)
