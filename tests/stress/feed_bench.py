"""Input feeding (row f1) at BASELINE config 2: one fused gather per batch from cines resident in HBM, timed with HIP
events on the launch stream, next to the oracle's per-sample CPU path (the reference's flip / crop / normalise / collate
on arrays that are ALREADY decoded - the reference additionally gunzips two cines per sample).
Usage: python tests/stress/feed_bench.py [N T U h]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd')
sys.path[:0] = [ROOT, PKG]
import numpy as np
import torch
from hipvsr.cine_cache import CineCache
from oracle import input_oracle as io_

N, T, U, h = (int(a) for a in (sys.argv[1:5] + [8, 7, 6, 128][len(sys.argv) - 1:]))
s, Tc, Hl = 4, 30, 160
dev = torch.device('cuda:0')
rng = np.random.RandomState(0)
cines = []
cache = CineCache(dev, s, [54.089], [48.084])
for i in range(16):
    hr = np.round(rng.rand(Hl * s, Hl * s, 1, Tc) * 255).astype(np.float32)
    lr = hr.reshape(Hl, s, Hl, s, 1, Tc).mean((1, 3)).astype(np.float32)
    code = np.cos(np.linspace(0, np.pi, Tc, endpoint=False))
    cines.append((lr, hr, code))
    cache.add_cine(lr, hr, code)
cache.finalize()
r = random.Random(0)
items = [(r.randrange(16), r.randrange(Tc)) for _ in range(N)]
draws = [cache.draw(c, (h, h), r) for c, _ in items]
for _ in range(3):
    b = cache.gather(items, draws, T, U, (h, h))
torch.cuda.synchronize()
reps = 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
for _ in range(reps):
    b = cache.gather(items, draws, T, U, (h, h))
e1.record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / reps
ms = e0.elapsed_time(e1) / reps
F = T + 2 * U
px = N * (F * h * h + T * (s * h) ** 2)
print(f'pool {cache.pool.numel() * 4 / 1e9:.2f} GB; batch N={N} T={T} F={F} {h}x{h} -> {s*h}x{s*h}: {px * 4 / 1e6:.1f} MB out')
print(f'gather: {ms * 1e3:.1f} us on the stream ({wall * 1e3:.3f} ms wall per call) = {px * 8 / ms / 1e6:.0f} GB/s algorithmic (8 B / pixel) of 8000 GB/s')
t0 = time.perf_counter()
want = io_.collate([io_.get_item(*cines[c], t, T, U, d, (h, h), s, [54.089], [48.084]) for (c, t), d in zip(items, draws)])
cpu = time.perf_counter() - t0
ok = all(np.array_equal(g.cpu().numpy(), w) for g, w in zip(b['lr_imgs'] + b['hr_imgs'], want[0] + want[1]))
print(f'CPU oracle (1 core, cines already decoded): {cpu * 1e3:.1f} ms per batch; bit-exact with the gather: {ok}')
print(f'supervised frames/s the feed sustains: {N * T / (ms / 1e3):.0f}')
