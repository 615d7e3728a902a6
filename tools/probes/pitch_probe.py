"""Probe: does the 272-byte pixel pitch of the refine block's 136-channel bf16 tensors (R1, dR1) cost the kernels that read them?
The same launches with the operand in a 128-channel tensor (256-byte pitch).  GPU box only."""
import os, sys
sys.argv = ['x', 'zzz']
__file__ = os.path.abspath('tools/kbench_bf16.py')
exec(open('tools/kbench_bf16.py').read().split("px = N * H * W")[0])
flt = ''
Hf, Hb, P8 = R(F * N, H, W, 64), R(F * N, H, W, 64), R(F * N, H, W, 8)
xs1 = []
for j in range(5):
    xs1 += [Src(Hf, img_off=(4 + j) * N), Src(Hb, img_off=(4 + j) * N), Src(P8, img_off=(4 + j) * N)]
xs1a = [sc for i, sc in enumerate(xs1) if i % 3 != 2] + [sc for i, sc in enumerate(xs1) if i % 3 == 2]
dw1, db1 = ops.empty(129, 645, 3, 3), ops.empty(129)
for C in (136, 128, 192):
    dR = R((T + 4) * N, H, W, C)
    timeit(f'refine1.wgrad.main dy C={C}', lambda: ops.wgrad(P.r1_wgrad_a, xs1a, [Src(dR, nch=128, img_off=2 * N)], TN, H, W, dw1, db1),
           2.0 * TN * H * W * 128 * 645 * 9, 1, 5)
# the LSTM weight gradient with its dy in a wider tensor (pitch 512 B vs 544 B)
pl = P.lstm[('forward', 1)]
xs, hs = R(TN + N, H, W, 64), R(TN + N, H, W, 64)
dw, db = ops.empty(256, 128, 3, 3), ops.empty(256)
for C in (256, 272):
    gd = R(TN, H, W, C)
    timeit(f'lstm.wgrad dy C={C}', lambda: ops.wgrad(pl['wgrad'], [Src(xs, img_off=N), Src(hs)], [Src(gd, nch=256)], TN, H, W, dw, db), 2.0 * TN * H * W * 128 * 256 * 9, 1, 5)
