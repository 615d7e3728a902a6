"""Whole-cycle testing of RefineNet (reference src/runner/predictors/acdc_vsr_refinenet_predictor.py:15-196): one cine
per batch (batch size 1 enforced, :23-24), ``net(inputs, pos_codes)[-1]`` under ``no_grad`` (:57-62), per-frame losses
(T, #losses) and metrics (T, #metrics) with the Cardiac* metrics receiving the patient name (:127-160), the running log
weighted by batch_size * T (:162-178), optional export of results.csv / frames / videos (:40-50, :67-95, :100-104).

Inside this boundary: up to ``cines_per_launch`` consecutive cines of one shape run through
the network as one batch (results per cine are unchanged), only the last output group is computed (``net.last_group_only``),
the forward is replayed from a HIP graph per cine shape (hipvsr.graph.GraphedForward; ``graph=False`` turns that off),
the per-frame losses are one fused launch and PSNR / SSIM of all frames - denormalisation included - another one.
Export differences: imageio / scipy.misc are not in this image, so frames are written by a built-in 8-bit PNG encoder
and the video of a slice is the (T, H, W) uint8 stack ``<sid>.npy`` instead of a GIF."""
import csv
import functools
import logging
import struct
import zlib
from pathlib import Path

import numpy as np
import torch
from tqdm import tqdm

from hipvsr.autograd import fused_losses
from hipvsr.graph import GraphedForward
from src.model.metrics import PSNR, SSIM, fused_metrics
from src.runner.predictors.base_predictor import BasePredictor
from src.utils import denormalize


def write_png_gray8(path, img):
    """Minimal 8-bit greyscale PNG (zlib from the standard library)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape

    def chunk(tag, data):
        body = tag + data
        return struct.pack('>I', len(data)) + body + struct.pack('>I', zlib.crc32(body) & 0xffffffff)
    raw = b''.join(b'\x00' + img[y].tobytes() for y in range(h))
    with open(path, 'wb') as f:
        f.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 0, 0, 0, 0)) +
                chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))


class AcdcVSRRefineNetPredictor(BasePredictor):
    def __init__(self, saved_dir=None, exported=False, graph=True, cines_per_launch=8, **kwargs):
        super().__init__(**kwargs)
        if self.test_dataloader.batch_size != 1:
            raise ValueError(f'The testing batch size should be 1. Got {self.test_dataloader.batch_size}.')
        if exported:
            self.saved_dir = Path(saved_dir)
        self.exported = exported
        self._denormalize = functools.partial(denormalize, dataset='acdc')
        self._graphed = GraphedForward(self.net) if graph else None
        # Consecutive cines of one shape (the slices of a patient) go through the network together as a batch: samples
        # are independent (bit for bit, quirk Q8), a batch-1 cell launch uses a tenth of the chip and takes as long as a
        # batch-8 one.  Losses, metrics, log and exports stay per cine, in loader order.  1 = one forward per cine.
        self.cines_per_launch = max(1, int(cines_per_launch))

    def _sample_name(self, index):
        """The file stem '<patient>_2d+1d_<sid>' of sample ``index`` (reference :59-61 reads it from dataset.data)."""
        i = int(index[0]) if torch.is_tensor(index) and index.dim() else int(index)
        cache = getattr(self.test_dataloader, 'cache', None)
        if cache is not None:
            return cache.names[self.test_dataloader.dataset.data[i][0]]
        data = getattr(self.test_dataloader.dataset, 'data', None)
        if data is not None and len(data) > i and isinstance(data[i], (tuple, list)) and hasattr(data[i][0], 'parts'):
            return data[i][0].parts[-1].split('.')[0]
        return f'patient{i:03d}_2d+1d_sequence01'

    def _forward(self, inputs, pos_codes):
        self.net.last_group_only = True            # the predictor consumes outputs[-1] only (:62)
        if self._graphed is not None:
            return self._graphed(inputs, pos_codes)
        return self.net(inputs, pos_codes)

    @staticmethod
    def _shape_key(item):
        inputs, targets, pos_codes, _ = item
        return len(inputs), tuple(inputs[0].shape), len(targets), tuple(targets[0].shape), tuple(pos_codes.shape)

    def predict(self):
        self.net.eval()
        trange = tqdm(self.test_dataloader, total=len(self.test_dataloader), desc='testing')
        results = [['name'] + [type(fn).__name__ for fn in self.metric_fns] + [type(fn).__name__ for fn in self.loss_fns]]
        state = dict(log=self._init_log(), count=0, results=results, bar=trange)
        pending = []
        for batch in trange:
            batch = self._allocate_data(batch)
            item = self._get_inputs_targets(batch)
            if pending and (len(pending) >= self.cines_per_launch or self._shape_key(item) != self._shape_key(pending[0])):
                self._run_group(pending, state)
                pending = []
            pending.append(item)
        if pending:
            self._run_group(pending, state)
        if self.exported:
            self.saved_dir.mkdir(parents=True, exist_ok=True)
            with open(self.saved_dir / 'results.csv', 'w', newline='') as f:
                csv.writer(f).writerows(results)
        log = state['log']
        for k in log:
            log[k] /= max(state['count'], 1)
        logging.info(f'Test log: {log}.')
        return log

    def _run_group(self, group, state):
        """K cines of one shape: one forward with N = K, then the reference's per-cine bookkeeping (:57-99)."""
        K = len(group)
        names = [self._sample_name(g[3]) for g in group]
        if K == 1:
            inputs, targets, pos_codes = group[0][:3]
        else:
            inputs = [torch.cat([g[0][k] for g in group]) for k in range(len(group[0][0]))]
            targets = [torch.cat([g[1][k] for g in group]) for k in range(len(group[0][1]))]
            pos_codes = torch.cat([g[2] for g in group])
        T = len(targets)
        with torch.no_grad():
            all_outputs = self._forward(inputs, pos_codes)
            outputs = all_outputs[-1]
            losses_all = self._compute_losses(outputs, targets, all_outputs)                 # (T, K, #loss_fns)
            metrics_all = self._compute_metrics(outputs, targets, [n.split('_')[0] for n in names], all_outputs)   # (T, K, #metric_fns)
            sr_all = None
            if self.exported:
                sr_all = torch.stack([self._denormalize(o) for o in outputs]).cpu().numpy().astype(np.uint8)   # (T, K, C, H, W)
        for i, filename in enumerate(names):
            patient, _, sid = filename.split('_')
            losses, metrics = losses_all[:, i], metrics_all[:, i]
            loss = (losses.mean(dim=0) * self.loss_weights).sum()
            if self.exported:
                stem = filename.replace('2d+1d', '2d').replace('sequence', 'slice')
                for t, (ls, ms) in enumerate(zip(losses.tolist(), metrics.tolist())):
                    state['results'].append([stem + f'_frame{t + 1:0>2d}', *ms, *ls])
                sr = sr_all[:, i].reshape(T, *sr_all.shape[-2:])
                (self.saved_dir / 'videos' / patient).mkdir(parents=True, exist_ok=True)
                np.save(self.saved_dir / 'videos' / patient / f'{sid}.npy', sr)
                (self.saved_dir / 'imgs' / patient).mkdir(parents=True, exist_ok=True)
                for t in range(T):
                    write_png_gray8(self.saved_dir / 'imgs' / patient / (sid.replace('sequence', 'slice') + f'_frame{t + 1:0>2d}.png'), sr[t])
            batch_size = self.test_dataloader.batch_size
            self._update_log(state['log'], batch_size, T, loss, losses, metrics)
            state['count'] += batch_size * T
        state['bar'].set_postfix(**{k: f'{v / state["count"]: .3f}' for k, v in state['log'].items()})

    def _get_inputs_targets(self, batch):
        return batch['lr_imgs'], batch['hr_imgs'], batch['pos_code'], batch['index']

    def _compute_losses(self, outputs, targets, all_outputs=None):
        """(T, N, #loss_fns): the loss of every frame of every cine on its own (reference :127-138 with N = 1)."""
        N = outputs[0].shape[0]
        cols = []
        for loss_fn in self.loss_fns:
            per = fused_losses(all_outputs, targets, loss_fn, last_only=True, per_sample=True) if all_outputs is not None else None
            if per is None:
                per = torch.stack([torch.stack([loss_fn(o[i:i + 1], t[i:i + 1]) for i in range(N)]) for o, t in zip(outputs, targets)])
            cols.append(per)
        return torch.stack(cols, dim=2)

    def _compute_metrics(self, outputs, targets, names, all_outputs=None):
        """(T, N, #metric_fns); ``names``: the patient of every sample (Cardiac* metrics crop by it, reference :140-160)."""
        N = outputs[0].shape[0]
        names = [names] * N if isinstance(names, str) else list(names)
        plain = [fn for fn in self.metric_fns if type(fn) in (PSNR, SSIM)]
        packed = getattr(all_outputs, 'packed', None)
        fused = fused_metrics(outputs, targets, plain, 'acdc', per_sample=True,
                              packed_last=packed[-1, -1].detach() if packed is not None else None) if plain else None
        den_o = den_t = None
        cols, k = [], 0
        for fn in self.metric_fns:
            if fused is not None and type(fn) in (PSNR, SSIM):
                cols.append(fused[:, :, k])
                k += 1
                continue
            if den_o is None:
                den_o, den_t = [self._denormalize(o) for o in outputs], [self._denormalize(t) for t in targets]
            if 'Cardiac' in type(fn).__name__:
                cols.append(torch.stack([torch.stack([fn(o[i:i + 1], t[i:i + 1], names[i]) for i in range(N)]) for o, t in zip(den_o, den_t)]))
            else:
                cols.append(torch.stack([torch.stack([fn(o[i:i + 1], t[i:i + 1]) for i in range(N)]) for o, t in zip(den_o, den_t)]))
        return torch.stack(cols, dim=2) if cols else torch.zeros(len(targets), N, 0, device=self.device)

    def _update_log(self, log, batch_size, T, loss, losses, metrics):
        log['Loss'] += loss.item() * batch_size * T
        for fn, v in zip(self.loss_fns, losses.mean(dim=0)):
            log[type(fn).__name__] += v.item() * batch_size * T
        for fn, v in zip(self.metric_fns, metrics.mean(dim=0)):
            log[type(fn).__name__] += v.item() * batch_size * T
