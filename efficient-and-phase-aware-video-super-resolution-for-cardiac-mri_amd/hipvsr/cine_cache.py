"""Input feeding with the cines resident in HBM.

The reference's ``AcdcVSRRefineNetDataset.__getitem__`` (src/data/datasets/acdc_vsr_refinenet_dataset.py:49-89)
decompresses two whole ``.nii.gz`` cines and unpickles the phase-code table for EVERY sample, then flips, crops and
normalises on the CPU; the default collate stacks N samples and the trainer copies the batch to the GPU.  Here every
cine is decoded once into one fp32 pool on the device (ACDC train: about 13 GB of 288 GB), and a batch is ONE launch of
``rnh_cine_gather`` (include/refinenet_hip.h) that writes the packed ``(F, N, h, w)`` / ``(T, N, sh, sw)`` / ``(N, F)``
buffers the engine consumes.  What the loader returns is the reference's batch contract (row A0): ``lr_imgs`` list[F] of
(N, 1, h, w), ``hr_imgs`` list[T] of (N, 1, sh, sw), ``pos_code`` (N, F, 1), ``index`` (N,) - as views of the packed
buffers, so the module and the fused loss take them without another copy.

Random draws follow the reference's order per sample (flip, flip, crop row, crop column with Python's ``random``,
transforms.py:344,371,443-444); which sample gets which draw differs from a multi-worker DataLoader by construction
(there each worker process owns a ``random`` state).  There is no CPU path: the gather needs the HIP library.
"""
import ctypes as C
import gzip
import math
import pickle
import random
import struct
from pathlib import Path

import numpy as np
import torch

from . import lib as L

_NIFTI_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16, 768: np.uint32}


def read_nifti(path):
    """Minimal NIfTI-1 single-file reader (``.nii`` / ``.nii.gz``) for the arrays ``acdc_preprocess.py:70-77`` writes
    with ``nib.save(nib.Nifti1Image(array, np.eye(4)))``: 348-byte header, voxels in Fortran order at ``vox_offset``.
    Returns the array in its on-disk dtype, like ``nib.load(p).get_data()`` does for unscaled data; scaled data
    (``scl_slope`` other than 0 / NaN / 1 with zero intercept) is refused instead of being silently mis-read."""
    raw = Path(path).read_bytes()
    if raw[:2] == b'\x1f\x8b':
        raw = gzip.decompress(raw)
    for end in '<>':
        if struct.unpack(end + 'i', raw[0:4])[0] == 348:
            break
    else:
        raise ValueError(f'{path}: not a NIfTI-1 file (sizeof_hdr != 348)')
    if raw[344:348] not in (b'n+1\0', b'ni1\0'):
        raise ValueError(f'{path}: bad NIfTI-1 magic {raw[344:348]!r}')
    dim = struct.unpack(end + '8h', raw[40:56])
    datatype, bitpix = struct.unpack(end + 'hh', raw[70:74])
    vox_offset, slope, inter = struct.unpack(end + 'fff', raw[108:120])
    if datatype not in _NIFTI_DTYPES:
        raise ValueError(f'{path}: unsupported NIfTI datatype {datatype}')
    if not (slope == 0 or math.isnan(slope) or (slope == 1 and inter == 0)):
        raise ValueError(f'{path}: scaled NIfTI data (slope {slope}, intercept {inter}) is not supported')
    shape = tuple(int(d) for d in dim[1:1 + dim[0]])
    dt = np.dtype(_NIFTI_DTYPES[datatype]).newbyteorder(end)
    n = int(np.prod(shape))
    off = int(vox_offset) if vox_offset >= 352 else 352
    return np.frombuffer(raw, dtype=dt, count=n, offset=off).reshape(shape, order='F').astype(dt.newbyteorder('='))


class CineCache:
    """All cines of one split, decoded once, resident on the device in one fp32 pool."""

    def __init__(self, device, downscale_factor, means=None, stds=None):
        if downscale_factor not in [2, 3, 4]:
            raise ValueError(f'The downscale factor should be 2, 3, 4. Got {downscale_factor}.')      # dataset :23-24
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise L.HipKernelError('CineCache needs a HIP device; there is no CPU gather in this package')
        self.lib = L.load()
        self.s = downscale_factor
        if (means is None) != (stds is None):
            raise ValueError('Both the means and the standard deviations should have values or None.')   # transforms.py:112-113
        if means is not None and (len(means) != 1 or len(stds) != 1):
            raise ValueError('CineCache serves single-channel cines (in_channels = 1)')
        self.normalize = means is not None
        # numpy's arithmetic on a float32 image with Python scalars: both constants are rounded to float32 first
        self.mean = float(np.float32(means[0])) if self.normalize else 0.0
        self.stdv = float(np.float32(stds[0] + 1e-10)) if self.normalize else 1.0                      # transforms.py:166
        self._host, self.table, self.names = [], [], []
        self._floats = 0
        self.pool = None

    # ---- filling -------------------------------------------------------------------------------------------
    def add_cine(self, lr, hr, code, name=None):
        """lr, hr: (H, W, 1, T) arrays as stored on disk; code: (T,) phase code of the patient."""
        lr, hr = np.asarray(lr), np.asarray(hr)
        if lr.ndim != 4 or hr.ndim != 4 or lr.shape[2] != 1 or hr.shape[2] != 1 or lr.shape[3] != hr.shape[3]:
            raise ValueError(f'cines must be (H, W, 1, T) with equal T; got {lr.shape} and {hr.shape}')
        if hr.shape[0] // lr.shape[0] != self.s or hr.shape[1] // lr.shape[1] != self.s:
            raise ValueError(f'The ratio between the HR images and the LR images should be {self.s}.')    # transforms.py:405-406
        Tc = lr.shape[3]
        code = np.asarray(code, dtype=np.float64).astype(np.float32).reshape(-1)                          # ToTensor's .float()
        if code.shape[0] != Tc:
            raise ValueError(f'phase code has {code.shape[0]} entries for a cine of {Tc} frames')
        parts = [np.ascontiguousarray(a[:, :, 0, :].transpose(2, 0, 1), dtype=np.float32).reshape(-1) for a in (lr, hr)] + [code]
        offs = []
        for p in parts:
            offs.append(self._floats)
            self._host.append(p)
            self._floats += p.size + (-p.size) % 4                 # 16-byte aligned parts
            if p.size % 4:
                self._host.append(np.zeros((-p.size) % 4, np.float32))
        self.table.append(dict(lr_off=offs[0], hr_off=offs[1], code_off=offs[2], Tc=Tc, Hl=lr.shape[0], Wl=lr.shape[1],
                               Hh=hr.shape[0], Wh=hr.shape[1]))
        self.names.append(name if name is not None else f'cine{len(self.table) - 1}')
        self.pool = None
        return len(self.table) - 1

    @classmethod
    def from_dir(cls, data_dir, type, downscale_factor, pos_code_path, device, means=None, stds=None):
        """The reference's directory layout (dataset :37-38): ``<data_dir>/<type>/LR/X<s>/**/*2d+1d*.nii.gz`` and
        ``<data_dir>/<type>/HR/**/*2d+1d*.nii.gz``; ``pos_code_path``: pickle of {patient: (T,) array} (:66-70)."""
        data_dir = Path(data_dir)
        lr_paths = sorted((data_dir / type / 'LR' / f'X{downscale_factor}').glob('**/*2d+1d*.nii.gz'))
        hr_paths = sorted((data_dir / type / 'HR').glob('**/*2d+1d*.nii.gz'))
        if not lr_paths or len(lr_paths) != len(hr_paths):
            raise FileNotFoundError(f'no cine pairs under {data_dir}/{type} (LR {len(lr_paths)}, HR {len(hr_paths)})')
        with open(pos_code_path, 'rb') as f:
            codes = pickle.load(f)
        self = cls(device, downscale_factor, means, stds)
        for lp, hp in zip(lr_paths, hr_paths):
            filename = lp.parts[-1].split('.')[0]
            patient = filename.split('_')[0]                                                              # :68-69
            self.add_cine(read_nifti(lp), read_nifti(hp), codes[patient], name=filename)
        self.finalize()
        return self

    def finalize(self):
        if self.pool is None:
            host = torch.from_numpy(np.concatenate(self._host)) if self._host else torch.zeros(0)
            self.pool = host.to(self.device)
        return self

    # ---- sample lists --------------------------------------------------------------------------------------
    def train_items(self):
        """(cine, target frame) for every frame of every cine (dataset :39-43)."""
        return [(c, t) for c, e in enumerate(self.table) for t in range(e['Tc'])]

    def draw(self, cine, size, rng=random, hflip_prob=0.5, vflip_prob=0.5):
        """(hflip, vflip, h0, w0) in the reference's order of draws (transforms.py:344, 371, 443-444)."""
        e = self.table[cine]
        hflip = rng.random() < hflip_prob
        vflip = rng.random() < vflip_prob
        if e['Hl'] - size[0] < 0 or e['Wl'] - size[1] < 0:
            raise ValueError(f"The image ({(e['Hl'], e['Wl'], 1)}) is smaller than the cropped size ({list(size)}). Please use a smaller cropped size.")
        return hflip, vflip, rng.randint(0, e['Hl'] - size[0]), rng.randint(0, e['Wl'] - size[1])

    # ---- one batch = one launch ----------------------------------------------------------------------------
    def gather(self, items, draws, num_frames, num_updated_frames, size=None, index=None):
        """items: list of (cine, t) - t the target frame of a training sample - or (cine, None) for the whole cycle
        (valid / test: no augmentation, whole frames; all cines of the batch must then agree in size and length).
        draws: per item (hflip, vflip, h0, w0) or None.  Returns the batch dict of row A0, on the device."""
        self.finalize()
        N = len(items)
        if N == 0:
            raise ValueError('empty batch')
        U = num_updated_frames
        e0 = self.table[items[0][0]]
        whole = items[0][1] is None
        if whole:
            T, F = e0['Tc'], e0['Tc'] + 2 * U
            h, w = e0['Hl'], e0['Wl']
            if U > e0['Tc']:
                raise ValueError(f'num_updated_frames ({U}) exceeds the cycle length ({e0["Tc"]})')
        else:
            T, F = num_frames, num_frames + 2 * U
            h, w = size
        desc = (L.CineSample * N)()
        for n, (cine, t) in enumerate(items):
            e = self.table[cine]
            d = desc[n]
            for k in ('lr_off', 'hr_off', 'code_off', 'Tc', 'Hl', 'Wl', 'Hh', 'Wh'):
                setattr(d, k, e[k])
            if (t is None) != whole:
                raise ValueError('a batch is either all training windows or all whole cycles')
            if whole:
                if (e['Tc'], e['Hl'], e['Wl']) != (e0['Tc'], e0['Hl'], e0['Wl']):
                    raise ValueError('whole-cycle batches need cines of one size and length (the reference uses batch size 1)')
                d.lr_start, d.hr_start = e['Tc'] - U, 0                                                    # dataset :86-88
            else:
                if not 0 <= t < e['Tc']:
                    raise IndexError(f'frame {t} of a cine with {e["Tc"]} frames')
                # the reference slices the tripled list [start-U : end+U] with start = t + Tc - T + 1 (:80-83); outside
                # 0 .. 3*Tc a Python slice would silently come back short, so that case is an error here
                if T + U - 1 > e['Tc'] or U > e['Tc']:
                    raise ValueError(f'a window of {T}+2*{U} frames does not fit the tripled cycle of {e["Tc"]} frames')
                d.hr_start = t + e['Tc'] - T + 1
                d.lr_start = d.hr_start - U
            dr = draws[n] if draws is not None else None
            if dr is not None and not whole:
                d.hflip, d.vflip, d.y0, d.x0 = int(bool(dr[0])), int(bool(dr[1])), int(dr[2]), int(dr[3])
        s = self.s
        inputs = torch.empty(F, N, 1, h, w, device=self.device, dtype=torch.float32)
        targets = torch.empty(T, N, 1, s * h, s * w, device=self.device, dtype=torch.float32)
        pos = torch.empty(N, F, 1, device=self.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            L.check(self.lib.rnh_cine_gather(self.pool.data_ptr(), self.pool.numel(), desc, None, N, F, T, s, h, w,
                                             int(self.normalize), self.mean, self.stdv, inputs.data_ptr(), targets.data_ptr(),
                                             pos.data_ptr(), stream), 'rnh_cine_gather')
        batch = {'lr_imgs': list(inputs.unbind(0)), 'hr_imgs': list(targets.unbind(0)), 'pos_code': pos,
                 'index': torch.as_tensor(index if index is not None else list(range(N)))}
        return batch


class _Items:
    """What the reference's loops read from ``dataloader.dataset``: ``len`` and ``.data`` (predictor :59)."""

    def __init__(self, data, type):
        self.data, self.type = data, type

    def __len__(self):
        return len(self.data)


class GpuCineLoader:
    """Iterates batches of a CineCache like the reference's ``Dataloader`` over ``AcdcVSRRefineNetDataset``: shuffled
    (cine, frame) samples with flips + crop for ``type='train'``, one whole cycle per batch otherwise.  Under
    torch.distributed every rank takes its own shard of the (identically shuffled) sample list, padded to equal length
    like ``DistributedSampler`` does."""

    def __init__(self, cache, type='train', batch_size=1, shuffle=False, num_frames=5, num_updated_frames=0, size=(32, 32),
                 flips=(True, True), seed=None, rank=None, world_size=None, drop_last=False):
        import torch.distributed as dist
        self.cache, self.type, self.batch_size, self.shuffle = cache, type, batch_size, shuffle
        self.T, self.U, self.size, self.flips = num_frames, num_updated_frames, tuple(size), flips
        # seed: src.main passes one derived from main.random_seed and the split (dataset kwarg ``loader_seed``).  Without one it is a
        # digest of the CURRENT state of Python's `random` - equal on all ranks and after a resume when the callers seeded alike -
        # read WITHOUT drawing from it: the reference's BaseTrainer takes its per-epoch numpy seeds from that generator
        # (base_trainer.py:49-54), and a draw here would shift them
        if seed is None:
            import zlib
            seed = zlib.crc32(repr((random.getstate()[1][:16], type)).encode()) & 0x7fffffff
        self.drop_last, self.seed, self.epoch = drop_last, seed, 0
        on = dist.is_available() and dist.is_initialized()
        self.rank = rank if rank is not None else (dist.get_rank() if on else 0)
        self.world = world_size if world_size is not None else (dist.get_world_size() if on else 1)
        if type == 'train':
            items = cache.train_items()
        else:
            if batch_size != 1:
                raise ValueError(f'The testing batch size should be 1. Got {batch_size}.')                # predictor :23-24
            items = [(c, None) for c in range(len(cache.table))]
        self.items = items
        self.dataset = _Items(items, type)
        self._reseed()

    def _reseed(self):
        # augmentation draws: per rank (like per-worker states) and per epoch
        self.rng = random.Random((self.seed * 1000003 + self.epoch) * 1009 + self.rank)

    def set_epoch(self, epoch):
        """Called by the trainer before every epoch (like DistributedSampler.set_epoch): order and draws of epoch e do not
        depend on how many epochs this object has iterated, so a resumed run continues the original sequence."""
        self.epoch = epoch
        self._reseed()

    def _order(self):
        n = len(self.items)
        order = list(range(n))
        if self.shuffle:
            random.Random(self.seed + self.epoch).shuffle(order)     # same permutation on every rank
        if self.world > 1:
            total = -(-n // self.world) * self.world
            order = (order + order[:total - n])[self.rank:total:self.world]
        return order

    def __len__(self):
        n = len(self._order())
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self):
        order = self._order()
        self.epoch += 1                                             # stand-alone use: a new permutation next time
        for b in range(0, len(order), self.batch_size):
            idx = order[b:b + self.batch_size]
            if self.drop_last and len(idx) < self.batch_size:
                return
            items = [self.items[i] for i in idx]
            draws = None
            if self.type == 'train':
                draws = [self.cache.draw(c, self.size, self.rng, 0.5 if self.flips[0] else 0.0, 0.5 if self.flips[1] else 0.0)
                         for c, _ in items]
            yield self.cache.gather(items, draws, self.T, self.U, self.size, index=idx)
