"""Training / validation step of RefineNet (reference src/runner/trainers/acdc_vsr_refinenet_trainer.py:17-136):
forward, deep-supervision loss (discount 0.5^(S-1-stage), mean over frames, sum over groups, :83-94; last group
only in evaluation, :95-100), zero_grad / backward / step (:41-47), PSNR-style metrics on the denormalised last
group (:103-120), running-mean log (:122-136).

Differences, all inside this boundary: L1 / Charbonnier over all 3*S*T (output, target) pairs go through ONE
fused HIP loss+gradient launch instead of 63 loss_fn calls; PSNR / SSIM of all frames of the step, denormalisation
included, are one launch (src.model.metrics.fused_metrics); and under torch.distributed the gradients are
averaged with one all-reduce before the optimizer step."""
import functools
import logging
import os

import numpy as np
import torch
from tqdm import tqdm

from hipvsr import dp
from hipvsr.autograd import discounted_total, fused_losses
from src.model.metrics import fused_metrics
from src.runner.trainers.base_trainer import BaseTrainer
from src.utils import denormalize


class AcdcVSRRefineNetTrainer(BaseTrainer):
    # graph: replay forward + loss + backward of a step from a HIP graph (hipvsr.graph.GraphedTrainStep).  OFF unless asked for
    # (trainer kwarg ``graph: true`` in the YAML, ``bench.py --graph on``): no measured shape gains from it - at the reference
    # YAML's 16 crops of 32 x 32 the step is GPU-bound either way (fp32 59.4 ms eager / 59.5 ms replayed, bf16 25.1 / 27.4:
    # tools/train_shape_bench.py, profiles/ARCHIVE/r02_e_train_shape.txt, r02_l_train_shape.txt) - and the graphed step behaves
    # differently (static gradient tensors, no zero_grad, one capture per batch shape).  Under torch.distributed the graph holds
    # forward + loss + backward only: the gradient all-reduce and the optimizer step stay outside it, as in the single-rank step
    # (the capture runs in thread-local error mode, so RCCL's watchdog thread cannot invalidate it;
    # tests/test_parity_r05.py::test_graphed_step_beside_a_live_rccl_communicator).

    def __init__(self, graph=False, **kwargs):
        super().__init__(**kwargs)
        self._denormalize = functools.partial(denormalize, dataset='acdc')
        self.graph = bool(graph)
        self._graphed = None

    def _total_loss(self, losses):
        """sum_i weight_i * losses[i] (trainer :45 of the reference).  One loss function of weight 1 - every reference YAML -
        is that loss itself: no stack / multiply / sum kernels around a scalar."""
        host = getattr(self, '_loss_weights_host', None)
        if host is None:                                                   # (read back once: no device sync per step, none in a graph capture)
            host = self._loss_weights_host = [float(v) for v in self.loss_weights.tolist()]
        if len(losses) == 1 and host[0] == 1.0:
            return losses[0]
        return (torch.stack(losses) * self.loss_weights).sum()

    def _backward(self, loss):
        """loss.backward() with a cached seed gradient (autograd otherwise fills a fresh ones_like per step)."""
        if loss.is_cuda:
            seed = getattr(self, '_seed_grad', None)
            if seed is None or seed.device != loss.device or seed.dtype != loss.dtype or seed.shape != loss.shape:
                seed = self._seed_grad = torch.ones_like(loss)
            loss.backward(gradient=seed)
        else:
            loss.backward()

    def _get_inputs_targets(self, batch):
        return batch['lr_imgs'], batch['hr_imgs'], batch['pos_code']

    def train_step(self, inputs, targets, pos_codes):
        """forward + loss + backward (+ gradient all-reduce) + optimizer step; returns (outputs, loss, losses)."""
        use_graph = bool(getattr(self, 'graph', False))
        if use_graph and dp.world() > 1 and os.environ.get('RNH_GRAPH_DP', '0') != '1':
            # a capture beside a live RCCL communicator has only ever run with ONE rank (tests/test_parity_r05.py) - no node with
            # more than one GPU has been available to this project: the watchdog with real peers, a first capture while another rank is still in
            # its all-reduce and a static gradient buffer handed to RCCL are untested.  Refused until someone opts in and measures it.
            raise RuntimeError('trainer kwarg graph=True under torch.distributed with more than one rank is untested on hardware: '
                               'set RNH_GRAPH_DP=1 to try it, or run the eager step (graph=False)')
        if use_graph:
            if getattr(self, '_graphed', None) is None:
                from hipvsr.graph import GraphedTrainStep
                self._graphed = GraphedTrainStep(self)
                logging.getLogger(__name__).info('training step replayed from a HIP graph (trainer kwarg graph=True)')
                # the flat Adam re-homes the parameters into its contiguous buffer on its first step: do that before the
                # first capture, which bakes their addresses in
                if hasattr(self.optimizer, '_adopt'):
                    for gi, group in enumerate(self.optimizer.param_groups):
                        if group['params']:
                            self.optimizer._adopt(gi, group)
            outputs, loss, losses = self._graphed(inputs, targets, pos_codes)      # gradients are overwritten, not accumulated
            dp.allreduce_gradients(self.net, force=bool(getattr(self, 'force_allreduce', False)))
            self.optimizer.step()
            return outputs, loss, losses
        outputs = self.net(inputs, pos_codes)
        losses = self._compute_losses(outputs, targets)
        loss = self._total_loss(losses)
        self.optimizer.zero_grad()
        self._backward(loss)
        dp.allreduce_gradients(self.net, force=bool(getattr(self, 'force_allreduce', False)))      # (force: the collective with ONE rank - tests)
        self.optimizer.step()
        return outputs, loss, losses

    def _run_epoch(self, mode):
        training = mode == 'training'
        self.net.train(training)
        loader = self.train_dataloader if training else self.valid_dataloader
        bar = tqdm(loader, total=len(loader), desc=mode)
        log, count = self._init_log(), 0
        batch = outputs = None
        for batch in bar:
            batch = self._allocate_data(batch)
            inputs, targets, pos_codes = self._get_inputs_targets(batch)
            T = len(inputs)
            if training:
                outputs, loss, losses = self.train_step(inputs, targets, pos_codes)
            else:
                with torch.no_grad():
                    outputs = self.net(inputs, pos_codes)
                    losses = self._compute_losses(outputs, targets)
                    loss = self._total_loss(losses)
            metrics = self._compute_metrics(outputs, targets)
            bs = loader.batch_size
            self._update_log(log, bs, T, loss, losses, metrics)
            count += bs * T
            bar.set_postfix(**{k: f'{v / count: .3f}' for k, v in log.items()})
        log, count = dp.allreduce_log(log, count, self.device)     # every rank sees the log of the WHOLE split
        for k in log:
            log[k] /= max(count, 1)
        return log, batch, (outputs[-1] if outputs is not None else None)

    def _compute_losses(self, outputs, targets):
        G, T = len(outputs), len(targets)
        losses = []
        for loss_fn in self.loss_fns:
            per_pair = fused_losses(outputs, targets, loss_fn)            # [G*T] or None
            if self.net.training:
                fused = discounted_total(outputs, per_pair, [np.power(0.5, (G // 3 - g // 3 - 1)) for g in range(G)], T)
                if fused is not None:                                      # the whole sum in one launch (hipvsr.autograd.LossTotalFn)
                    losses.append(fused)
                    continue
                terms = []
                for g in range(G):
                    discount = np.power(0.5, (G // 3 - g // 3 - 1))
                    if per_pair is not None:
                        vals = per_pair[g * T:(g + 1) * T] * discount
                    else:
                        vals = torch.stack([loss_fn(o, t) * discount for o, t in zip(outputs[g], targets)])
                    terms.append(vals.mean())
                losses.append(torch.stack(terms).sum())
            else:
                if per_pair is not None:
                    losses.append(per_pair[(G - 1) * T:].mean())
                else:
                    losses.append(torch.stack([loss_fn(o, t) for o, t in zip(outputs[-1], targets)]).mean())
        return losses

    def _compute_metrics(self, outputs, targets):
        packed = getattr(outputs, 'packed', None)
        fused = fused_metrics(outputs[-1], targets, self.metric_fns, 'acdc',
                              packed_last=packed[-1, -1].detach() if packed is not None else None)
        if fused is not None:
            return fused
        outs = [self._denormalize(o) for o in outputs[-1]]
        tgts = [self._denormalize(t) for t in targets]
        return [torch.stack([fn(o, t) for o, t in zip(outs, tgts)]).mean() for fn in self.metric_fns]

    def _update_log(self, log, batch_size, T, loss, losses, metrics):
        log['Loss'] += loss.item() * batch_size * T
        for fn, v in zip(self.loss_fns, losses):
            log[type(fn).__name__] += v.item() * batch_size * T
        for fn, v in zip(self.metric_fns, metrics):
            log[type(fn).__name__] += v.item() * batch_size * T
