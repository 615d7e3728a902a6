"""Input feeding: oracle/input_oracle.py against the reference's own transforms
(tests/golden/g6_input.pt), the host logic of hipvsr/cine_cache.py without a GPU, and - on the MI355X - the
``rnh_cine_gather`` launch against the oracle, bit for bit (pure data movement plus one IEEE subtract / divide)."""
import ctypes
import gzip
import os
import pickle
import random
import struct

import numpy as np
import pytest
import torch

from oracle import input_oracle as io_
from hipvsr import cine_cache as cc
from hipvsr import lib as L


@pytest.fixture(scope='module')
def g6(golden_dir):
    return torch.load(os.path.join(golden_dir, 'g6_input.pt'), weights_only=False)


def _frames(a):
    return [a[..., t] for t in range(a.shape[-1])]


def test_oracle_transforms_match_the_reference_classes(g6):
    """Same seed => same flips and crop as the reference's Compose([HFlip, VFlip, CropPatch]) and the same normalised
    tensors as Compose([Normalize, ToTensor]) (+ permute), bit for bit."""
    flips = 0
    for case in g6:
        lr, hr, s, size = case['lr'].numpy(), case['hr'].numpy(), case['s'], case['size']
        for run in case['runs']:
            draws = io_.draw_augment(random.Random(run['seed']), lr.shape, size)
            flips += int(draws[0]) + int(draws[1])
            aug = io_.augment(_frames(lr) + _frames(hr), *draws, size, s)
            assert len(aug) == len(run['aug'])
            for a, b in zip(aug, run['aug']):
                assert np.array_equal(a, b.numpy())
            out = [np.ascontiguousarray(io_.normalize(a, [54.089], [48.084]).transpose(2, 0, 1)) for a in aug]
            for a, b in zip(out, run['out']):
                assert a.dtype == np.float32 and np.array_equal(a, b.numpy())
        code = case['code'].numpy()
        assert torch.equal(torch.from_numpy(code.astype(np.float32)), case['pos_code'])
        whole = io_.get_item(lr, hr, code, None, 0, 0, means=[54.089], stds=[48.084])
        Tc = lr.shape[-1]
        for a, b in zip(whole[0] + whole[1], case['whole'][:Tc] + case['whole'][Tc:]):
            assert np.array_equal(a, b.numpy())
    assert flips > 10                                                # both kinds of flips were exercised


def test_oracle_window_arithmetic():
    """dataset :74-88 on a cine whose frames carry their own index."""
    Tc, T, U = 6, 3, 2
    lr = np.arange(Tc, dtype=np.float32).reshape(1, 1, 1, Tc) * np.ones((4, 4, 1, 1), np.float32)
    hr = np.arange(Tc, dtype=np.float32).reshape(1, 1, 1, Tc) * np.ones((8, 8, 1, 1), np.float32) + 100
    code = np.arange(Tc) / 10
    for t in range(Tc):
        l, h, c = io_.get_item(lr, hr, code, t, T, U)
        assert [int(x[0, 0, 0]) for x in l] == [(t - T + 1 - U + k) % Tc for k in range(T + 2 * U)]
        assert [int(x[0, 0, 0]) - 100 for x in h] == [(t - T + 1 + i) % Tc for i in range(T)]
        assert np.allclose(c[:, 0] * 10, [(t - T + 1 - U + k) % Tc for k in range(T + 2 * U)])
    l, h, c = io_.get_item(lr, hr, code, None, T, U)
    assert [int(x[0, 0, 0]) for x in l] == [(k - U) % Tc for k in range(Tc + 2 * U)] and len(h) == Tc and c.shape == (Tc + 2 * U, 1)


def write_nifti(path, arr):
    """NIfTI-1 single file by the published layout (what nib.save(nib.Nifti1Image(arr, np.eye(4))) produces for a
    float32 / int16 array): 348-byte header, 4 bytes of extension flags, voxels in Fortran order."""
    codes = {np.dtype(np.float32): (16, 32), np.dtype(np.int16): (4, 16), np.dtype(np.uint8): (2, 8)}
    dt, bits = codes[arr.dtype]
    hdr = bytearray(348)
    struct.pack_into('<i', hdr, 0, 348)
    struct.pack_into('<8h', hdr, 40, arr.ndim, *(list(arr.shape) + [1] * (7 - arr.ndim)))
    struct.pack_into('<hh', hdr, 70, dt, bits)
    struct.pack_into('<8f', hdr, 76, 1, 1, 1, 1, 1, 1, 1, 1)
    struct.pack_into('<fff', hdr, 108, 352.0, float('nan'), float('nan'))
    hdr[344:348] = b'n+1\0'
    blob = bytes(hdr) + b'\0\0\0\0' + arr.tobytes(order='F')
    with open(path, 'wb') as f:
        f.write(gzip.compress(blob) if str(path).endswith('.gz') else blob)


def test_read_nifti_roundtrip_and_errors(tmp_path):
    rng = np.random.RandomState(0)
    for arr in [rng.rand(7, 5, 1, 4).astype(np.float32), rng.randint(-5, 900, (6, 4, 3, 2)).astype(np.int16)]:
        for name in ('a.nii.gz', 'a.nii'):
            write_nifti(tmp_path / name, arr)
            got = cc.read_nifti(tmp_path / name)
            assert got.dtype == arr.dtype and got.shape == arr.shape and np.array_equal(got, arr)
    (tmp_path / 'bad.nii').write_bytes(b'\0' * 400)
    with pytest.raises(ValueError, match='NIfTI'):
        cc.read_nifti(tmp_path / 'bad.nii')
    blob = bytearray(gzip.decompress((tmp_path / 'a.nii.gz').read_bytes()))
    struct.pack_into('<ff', blob, 112, 2.0, 1.0)                     # scaled data: refused, not mis-read
    (tmp_path / 'scaled.nii').write_bytes(bytes(blob))
    with pytest.raises(ValueError, match='scaled'):
        cc.read_nifti(tmp_path / 'scaled.nii')


def test_cache_has_no_cpu_path_and_binding_matches_header():
    with pytest.raises(L.HipKernelError):
        cc.CineCache('cpu', 4)
    assert ctypes.sizeof(L.CineSample) == 80
    lib = L.load()
    d = (L.CineSample * 1)()
    assert lib.rnh_cine_gather(None, 0, d, None, 1, 1, 1, 1, 1, 1, 0, 0.0, 1.0, None, None, None, None) != 0
    assert b'null' in lib.rnh_last_error()


class _FakeCache:
    def __init__(self, lens):
        self.table = [dict(Tc=t, Hl=16, Wl=16) for t in lens]

    train_items = cc.CineCache.train_items
    draw = cc.CineCache.draw


def test_packed_view_needs_one_storage():
    """Adjacent addresses are not enough: separately allocated tensors that happen to be neighbours (the caching
    allocator does that) must go through a copy; slices of one buffer must not."""
    from hipvsr.hip_ops import packed_view
    buf = torch.arange(24, dtype=torch.float32)
    views = [buf[6 * k:6 * k + 6].view(2, 3) for k in range(4)]
    assert packed_view(views).data_ptr() == buf.data_ptr()
    raw = bytearray(buf.numpy().tobytes())
    apart = [torch.frombuffer(raw, dtype=torch.float32, count=6, offset=24 * k).view(2, 3) for k in range(4)]
    assert apart[1].data_ptr() == apart[0].data_ptr() + 24
    got = packed_view(apart)
    assert got.data_ptr() != apart[0].data_ptr() and torch.equal(got, buf.view(4, 2, 3))


def test_loader_shards_like_a_distributed_sampler():
    cache = _FakeCache([5, 7, 4])
    full = cc.GpuCineLoader(cache, 'train', batch_size=4, shuffle=True, seed=3, rank=0, world_size=1)
    assert len(full.items) == 16 and len(full) == 4
    seen = []
    for r in range(3):
        ld = cc.GpuCineLoader(cache, 'train', batch_size=4, shuffle=True, seed=3, rank=r, world_size=3)
        o = ld._order()
        assert len(o) == 6                                           # ceil(16 / 3): padded with the head of the list
        seen += o
    assert sorted(set(seen)) == list(range(16)) and len(seen) == 18
    assert cc.GpuCineLoader(cache, 'train', batch_size=4, shuffle=True, seed=3, rank=1, world_size=3)._order() == \
        cc.GpuCineLoader(cache, 'train', batch_size=4, shuffle=True, seed=3, rank=1, world_size=3)._order()
    with pytest.raises(ValueError, match='batch size should be 1'):
        cc.GpuCineLoader(cache, 'valid', batch_size=2)
    # draws in the reference's order
    assert cache.draw(0, (8, 8), random.Random(11)) == io_.draw_augment(random.Random(11), (16, 16, 1), (8, 8))
    with pytest.raises(ValueError, match='smaller than the cropped size'):
        cache.draw(0, (32, 8), random.Random(0))


# ---------------------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------------------
def _cines(rng, s, n, tc_range=(7, 12)):
    out = []
    for i in range(n):
        Hl, Wl, Tc = rng.randint(20, 40), rng.randint(20, 40), rng.randint(*tc_range)
        hr = np.round(rng.rand(Hl * s, Wl * s, 1, Tc) * 255).astype(np.float32)
        lr = (rng.rand(Hl, Wl, 1, Tc) * 255).astype(np.float32)
        out.append((lr, hr, np.cos(np.linspace(0, np.pi, Tc, endpoint=False))))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize('s', [2, 3, 4])
def test_gather_matches_oracle_bit_for_bit(s):
    dev = torch.device('cuda:0')
    rng = np.random.RandomState(s)
    cines = _cines(rng, s, 5)
    cache = cc.CineCache(dev, s, [54.089], [48.084])
    for lr, hr, code in cines:
        cache.add_cine(lr, hr, code)
    T, U, size = 3, 2, (16, 19)
    r = random.Random(5)
    items = [(r.randrange(len(cines)), None) for _ in range(9)]
    items = [(c, r.randrange(cines[c][0].shape[-1])) for c, _ in items]
    draws = [cache.draw(c, size, r) for c, _ in items]
    assert any(d[0] for d in draws) and any(d[1] for d in draws) and not all(d[0] for d in draws)
    batch = cache.gather(items, draws, T, U, size)
    torch.cuda.synchronize()
    want = io_.collate([io_.get_item(*cines[c], t, T, U, d, size, s, [54.089], [48.084]) for (c, t), d in zip(items, draws)])
    assert len(batch['lr_imgs']) == T + 2 * U and len(batch['hr_imgs']) == T
    for got, ref in zip(batch['lr_imgs'] + batch['hr_imgs'], want[0] + want[1]):
        assert got.shape == ref.shape and np.array_equal(got.cpu().numpy(), ref)
    assert np.array_equal(batch['pos_code'].cpu().numpy(), want[2])
    # the lists are views of the packed buffers: the module and the fused loss take them without a copy
    from hipvsr.hip_ops import packed_view
    assert packed_view(batch['lr_imgs']).data_ptr() == batch['lr_imgs'][0].data_ptr()
    assert packed_view(batch['hr_imgs']).data_ptr() == batch['hr_imgs'][0].data_ptr()
    # whole cycle (valid / test), no augmentation, unnormalised cache
    raw = cc.CineCache(dev, s)
    for lr, hr, code in cines:
        raw.add_cine(lr, hr, code)
    b = raw.gather([(2, None)], None, T, U)
    w = io_.collate([io_.get_item(*cines[2], None, T, U)])
    for got, ref in zip(b['lr_imgs'] + b['hr_imgs'], w[0] + w[1]):
        assert np.array_equal(got.cpu().numpy(), ref)
    assert np.array_equal(b['pos_code'].cpu().numpy(), w[2])
    # a crop outside the image is refused on the host, before any launch
    with pytest.raises(L.HipKernelError, match='outside'):
        cache.gather([items[0]], [(False, False, 1000, 0)], T, U, size)
    with pytest.raises(ValueError, match='does not fit'):
        cache.gather([items[0]], [draws[0]], 9, 6, size)


@pytest.mark.gpu
def test_back_to_back_gathers_keep_their_own_descriptors():
    """120 gathers of DIFFERENT samples enqueued back to back behind a busy stream, no synchronisation in between, the
    host descriptor arrays freed and re-filled as Python pleases; batches of 40 samples span two launches (32 descriptors
    per launch).  Every batch must equal the oracle's for ITS items.  (Round 1 uploaded the descriptors with an
    asynchronous copy from the caller's pageable array, which the next gather re-used - safe only because the runtime
    stages such copies before returning; the descriptors now travel as kernel arguments.)"""
    dev = torch.device('cuda:0')
    s = 2
    rng = np.random.RandomState(77)
    cines = _cines(rng, s, 6)
    cache = cc.CineCache(dev, s, [54.089], [48.084])
    for lr, hr, code in cines:
        cache.add_cine(lr, hr, code)
    T, U, size = 2, 2, (16, 16)
    r = random.Random(9)
    busy = torch.randn(4096, 4096, device=dev)
    for _ in range(6):
        busy = busy @ busy * 1e-3                                            # keeps the stream busy while the host runs ahead
    rounds = []
    for k in range(120):
        n = 40 if k % 10 == 0 else 1 + k % 3
        items = [(r.randrange(len(cines)), None) for _ in range(n)]
        items = [(c, r.randrange(cines[c][0].shape[-1])) for c, _ in items]
        draws = [cache.draw(c, size, r) for c, _ in items]
        rounds.append((items, draws, cache.gather(items, draws, T, U, size)))
    torch.cuda.synchronize()
    for items, draws, batch in rounds:
        want = io_.collate([io_.get_item(*cines[c], t, T, U, d, size, s, [54.089], [48.084]) for (c, t), d in zip(items, draws)])
        for got, ref in zip(batch['lr_imgs'] + batch['hr_imgs'], want[0] + want[1]):
            assert np.array_equal(got.cpu().numpy(), ref)
        assert np.array_equal(batch['pos_code'].cpu().numpy(), want[2])


@pytest.mark.gpu
def test_loader_feeds_a_training_step_from_nifti_files(tmp_path):
    """Reference directory layout on disk -> CineCache.from_dir -> GpuCineLoader -> RefineNet step."""
    s, rng = 4, np.random.RandomState(1)
    codes = {}
    for p in range(2):
        Tc = 19 + p
        codes[f'patient{p:03d}'] = np.cos(np.linspace(0, np.pi, Tc, endpoint=False))
        for q in range(2):
            lr, hr, _ = _cines(rng, s, 1, (Tc, Tc + 1))[0]
            name = f'patient{p:03d}_2d+1d_sequence{q + 1:02d}.nii.gz'
            for sub, a in ((f'LR/X{s}', lr), ('HR', hr)):
                d = tmp_path / 'train' / sub / f'patient{p:03d}'
                d.mkdir(parents=True, exist_ok=True)
                write_nifti(d / name, a)
    with open(tmp_path / 'position_code.pkl', 'wb') as f:
        pickle.dump(codes, f)
    dev = torch.device('cuda:0')
    cache = cc.CineCache.from_dir(tmp_path, 'train', s, tmp_path / 'position_code.pkl', dev, [54.089], [48.084])
    assert len(cache.table) == 4
    loader = cc.GpuCineLoader(cache, 'train', batch_size=3, shuffle=True, num_frames=3, num_updated_frames=6, size=(16, 16), seed=1)
    from src.model.nets import RefineNet
    torch.manual_seed(0)
    net = RefineNet(1, 1, [8, 8], num_stages=2, upscale_factor=4, update_memory=True, num_updated_frames=6, positional_encoding=True).to(dev)
    batch = next(iter(loader))
    outs = net(batch['lr_imgs'], batch['pos_code'])
    loss = sum(torch.nn.functional.l1_loss(o, t) for o, t in zip(outs[-1], batch['hr_imgs']))
    loss.backward()
    assert torch.isfinite(loss) and all(p.grad is None or torch.isfinite(p.grad).all() for p in net.parameters())
    assert len(loader) == -(-len(cache.train_items()) // 3)


@pytest.mark.gpu
def test_src_main_trains_from_cines_on_disk(tmp_path):
    """python -m src.main <yaml> with the reference's dataset section pointing at .nii.gz cines: served from HBM."""
    import types
    import yaml
    from conftest import PKG
    from src import main as M
    s, rng = 4, np.random.RandomState(2)
    codes = {}
    for split, npat in (('train', 2), ('valid', 1)):
        for p in range(npat):
            pid = f'patient{(100 if split == "valid" else 0) + p:03d}'
            Tc = 14 + p
            codes[pid] = np.cos(np.linspace(0, np.pi, Tc, endpoint=False))
            Hl, Wl = 36 + 4 * p, 40
            hr = np.round(rng.rand(Hl * s, Wl * s, 1, Tc) * 255).astype(np.float32)
            lr = hr.reshape(Hl, s, Wl, s, 1, Tc).mean((1, 3)).astype(np.float32)
            for sub, a in ((f'LR/X{s}', lr), ('HR', hr)):
                d = tmp_path / 'data' / split / sub / pid
                d.mkdir(parents=True, exist_ok=True)
                write_nifti(d / f'{pid}_2d+1d_sequence01.nii.gz', a)
    with open(tmp_path / 'position_code.pkl', 'wb') as f:
        pickle.dump(codes, f)
    cfg = yaml.safe_load(open(os.path.join(PKG, 'configs', 'refine_net_x4_synthetic.yaml')))
    cfg['main']['saved_dir'] = str(tmp_path / 'run')
    cfg['dataset']['kwargs'].update(data_dir=str(tmp_path / 'data'), pos_code_path=str(tmp_path / 'position_code.pkl'))
    cfg['trainer']['kwargs']['num_epochs'] = 1
    cfg['dataloader']['kwargs'].update(num_workers=0, train_batch_size=4)
    cfg['monitor']['kwargs']['saved_freq'] = 1
    p = tmp_path / 'cfg.yaml'
    p.write_text(yaml.safe_dump(cfg))
    M.main(types.SimpleNamespace(config_path=p, test=False))
    assert (tmp_path / 'run' / 'checkpoints' / 'model_1.pth').exists()
    log = (tmp_path / 'run' / 'log' / 'scalars.jsonl').read_text()
    assert '"Loss"' in log and '"PSNR"' in log
