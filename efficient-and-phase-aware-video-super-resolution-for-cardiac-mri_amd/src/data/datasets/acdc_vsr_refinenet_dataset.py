"""Input contract of the hot path: every sample is
``{'lr_imgs': list[F] of (1,h,w), 'hr_imgs': list[T] of (1,sh,sw), 'pos_code': (F,1), 'index': i}`` with
F = num_frames + 2*num_updated_frames, frames cut from the cyclically tripled cine and the phase code sliced the
same way (reference src/data/datasets/acdc_vsr_refinenet_dataset.py:49-89).

``AcdcVSRRefineNetDataset`` accepts the reference's constructor kwargs.  Cines found under ``data_dir`` (the
reference's layout of .nii.gz files) are decoded once and served from HBM (hipvsr/cine_cache.py, row f1); when
``data_dir`` holds no data (the ACDC files are not part of this offline image) it serves a deterministic synthetic
cine with the statistics of the normalised data.
"""
import math
from pathlib import Path

import torch
from torch.utils.data import Dataset


class SyntheticCineDataset(Dataset):
    """Blurred-noise background + a disc whose radius follows the cardiac phase; LR = average pooling of HR."""

    def __init__(self, downscale_factor=4, num_frames=7, num_updated_frames=6, size=(32, 32), cycle=30, length=64,
                 seed=20200526, type='train', **_):
        self.s, self.T, self.U = downscale_factor, num_frames, num_updated_frames
        self.size, self.cycle, self.length, self.seed, self.type = tuple(size), cycle, length, seed, type

    def __len__(self):
        return self.length

    def _cine(self, index):
        g = torch.Generator('cpu').manual_seed(self.seed + index)
        h, w = self.size[0] * self.s, self.size[1] * self.s
        base = torch.randn(1, 1, h, w, generator=g)
        k = torch.arange(-6, 7, dtype=torch.float32)
        k = torch.exp(-(k / 3.0) ** 2 / 2)
        k = (k / k.sum()).view(1, 1, -1, 1)
        base = torch.nn.functional.conv2d(torch.nn.functional.conv2d(base, k, padding=(6, 0)), k.transpose(2, 3), padding=(0, 6))
        base = (base - base.mean()) / (base.std() + 1e-6)
        yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
        phi = int(torch.randint(0, self.cycle, (1,), generator=g))
        frames, codes = [], []
        for t in range(self.cycle):
            ph = math.cos(2 * math.pi * (t + phi) / self.cycle)
            rad = 0.25 * min(h, w) * (1 + 0.3 * ph)
            disc = ((yy - h / 2) ** 2 + (xx - w / 2) ** 2 < rad ** 2).float()
            frames.append((base[0] * 0.5 + disc * 1.5).clamp(-1.2, 3.5))
            codes.append(ph)
        return frames, torch.tensor(codes, dtype=torch.float32)

    def __getitem__(self, index):
        hr, code = self._cine(index)
        lr = [torch.nn.functional.avg_pool2d(f.unsqueeze(0), self.s)[0] for f in hr]
        Tc = len(hr)
        lr3, hr3, code3 = lr * 3, hr * 3, code.repeat(3).unsqueeze(1)
        if self.type == 'train':
            t = (index % Tc) + Tc
            start, end = t - self.T + 1, t + 1
            return {'lr_imgs': lr3[start - self.U:end + self.U], 'hr_imgs': hr3[start:end],
                    'pos_code': code3[start - self.U:end + self.U], 'index': index}
        return {'lr_imgs': lr3[Tc - self.U:2 * Tc + self.U], 'hr_imgs': hr3[:Tc], 'pos_code': code3[Tc - self.U:2 * Tc + self.U],
                'index': index}


class AcdcVSRRefineNetDataset(SyntheticCineDataset):
    """Reference constructor kwargs (dataset :21-36).  With cines on disk (the reference's directory layout) the
    samples are served from HBM by ``hipvsr.cine_cache`` - ``src.data.dataloader.Dataloader`` turns this dataset into a
    ``GpuCineLoader`` - and ``transforms`` / ``augments`` are read as parameters of the fused gather: Normalize's
    means / stds, the two flips, RandomCropPatch's size (ToTensor is implied).  Anything else in those lists is refused
    rather than ignored."""

    def __init__(self, downscale_factor, transforms=None, pos_code_path=None, augments=None, num_frames=5,
                 num_updated_frames=0, data_dir=None, type='train', device=None, loader_seed=None, **kwargs):
        if downscale_factor not in [2, 3, 4]:
            raise ValueError(f'The downscale factor should be 2, 3, 4. Got {downscale_factor}.')
        size, flips, means, stds = (32, 32), [False, False], None, None
        for a in (augments or []):
            a = dict(a)
            kw = dict(a.get('kwargs') or {})
            if a.get('name') == 'RandomCropPatch':
                size = tuple(kw.get('size', size))
                if kw.get('ratio', downscale_factor) != downscale_factor:
                    raise ValueError(f"The ratio between the HR images and the LR images should be {kw.get('ratio')}.")
            elif a.get('name') == 'RandomHorizontalFlip' and not kw:
                flips[0] = True
            elif a.get('name') == 'RandomVerticalFlip' and not kw:
                flips[1] = True
            else:
                raise ValueError(f"augmentation {a.get('name')} {kw} is not part of the fused input path")
        for tr in (transforms or []):
            tr = dict(tr)
            kw = dict(tr.get('kwargs') or {})
            if tr.get('name') == 'Normalize':
                means, stds = kw.get('means'), kw.get('stds')
            elif tr.get('name') != 'ToTensor':
                raise ValueError(f"transform {tr.get('name')} is not part of the fused input path")
        self.cache = None
        if data_dir is not None and any(Path(data_dir).glob(f'{type}/**/*2d+1d*.nii.gz')):
            from hipvsr.cine_cache import CineCache
            if device is None:
                raise ValueError('cines on disk are served from HBM: pass device= (src.main does)')
            self.cache = CineCache.from_dir(data_dir, type, downscale_factor, pos_code_path, device, means, stds)
            self.loader_kwargs = dict(type=type, num_frames=num_frames, num_updated_frames=num_updated_frames, size=size, flips=tuple(flips))
            if loader_seed is not None:      # src.main derives it from main.random_seed: the loaders draw nothing from the global RNG
                import zlib
                self.loader_kwargs['seed'] = zlib.crc32(f'{loader_seed}:{type}'.encode()) & 0x7fffffff
            self.type = type
            self.data = self.cache.train_items() if type == 'train' else [(c, None) for c in range(len(self.cache.table))]
            return
        super().__init__(downscale_factor=downscale_factor, num_frames=num_frames, num_updated_frames=num_updated_frames,
                         size=size if type == 'train' else (54, 64), length=64 if type == 'train' else 2, type=type)

    def __len__(self):
        return len(self.data) if self.cache is not None else super().__len__()

    def __getitem__(self, index):
        if self.cache is not None:
            raise RuntimeError('this dataset lives in HBM: iterate it through src.data.dataloader.Dataloader (one fused gather per batch)')
        return super().__getitem__(index)
