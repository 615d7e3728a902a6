"""Input contract of the hot path: every sample is
``{'lr_imgs': list[F] of (1,h,w), 'hr_imgs': list[T] of (1,sh,sw), 'pos_code': (F,1), 'index': i}`` with
F = num_frames + 2*num_updated_frames, frames cut from the cyclically tripled cine and the phase code sliced the
same way (reference src/data/datasets/acdc_vsr_refinenet_dataset.py:49-89).

The ACDC NIfTI files and nibabel are not available offline, so ``AcdcVSRRefineNetDataset`` accepts the
reference's constructor kwargs and, when ``data_dir`` holds no data, serves a deterministic synthetic cine with
the statistics of the normalised data.  Feeding real data at GPU speed is the "next" row f1.
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
    def __init__(self, downscale_factor, transforms=None, pos_code_path=None, augments=None, num_frames=5,
                 num_updated_frames=0, data_dir=None, type='train', **kwargs):
        if downscale_factor not in [2, 3, 4]:
            raise ValueError(f'The downscale factor should be 2, 3, 4. Got {downscale_factor}.')
        size = (32, 32)
        for a in (augments or []):
            if dict(a).get('name') == 'RandomCropPatch':
                size = tuple(dict(a).get('kwargs', {}).get('size', size))
        if data_dir is not None and any(Path(data_dir).glob('**/*2d+1d*.nii.gz')):
            raise NotImplementedError('NIfTI loading needs nibabel, which is not part of this offline image; '
                                      'convert the cines to tensors or install nibabel (next-row f1).')
        super().__init__(downscale_factor=downscale_factor, num_frames=num_frames, num_updated_frames=num_updated_frames,
                         size=size if type == 'train' else (54, 64), length=64 if type == 'train' else 2, type=type)
