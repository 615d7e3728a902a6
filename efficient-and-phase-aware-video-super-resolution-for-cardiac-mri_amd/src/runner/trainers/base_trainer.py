"""Epoch loop, checkpointing and logging shell around the hot path (reference
src/runner/trainers/base_trainer.py:46-97, :146-161, :224-252).  Host logic only - nothing here is accelerated.
Checkpoints keep the reference's keys ('net', 'optimizer', 'lr_scheduler', 'monitor', 'epoch', 'random_state',
'np_random_seeds') so that either code base can resume the other's files."""
import logging
import random

import numpy as np
import torch

from hipvsr import dp


class BaseTrainer:
    def __init__(self, device, train_dataloader, valid_dataloader, net, loss_fns, loss_weights, metric_fns, optimizer,
                 lr_scheduler, logger, monitor, num_epochs):
        self.device = device
        self.train_dataloader, self.valid_dataloader = train_dataloader, valid_dataloader
        self.net = net.to(device)
        self.loss_fns = [fn.to(device) for fn in loss_fns]
        self.loss_weights = torch.tensor(loss_weights, dtype=torch.float, device=device)
        self.metric_fns = [fn.to(device) for fn in metric_fns]
        self.optimizer = optimizer
        if isinstance(lr_scheduler, torch.optim.lr_scheduler.CyclicLR):
            raise NotImplementedError('Do not support torch.optim.lr_scheduler.CyclicLR scheduler yet.')
        self.lr_scheduler = lr_scheduler
        self.logger, self.monitor = logger, monitor
        self.num_epochs = num_epochs
        self.epoch = 1
        self.np_random_seeds = None
        dp.broadcast_parameters(self.net)

    def train(self):
        if self.np_random_seeds is None:
            self.np_random_seeds = random.sample(range(10000000), k=self.num_epochs)
        while self.epoch <= self.num_epochs:
            np.random.seed(self.np_random_seeds[self.epoch - 1])
            for loader in (self.train_dataloader, self.valid_dataloader):
                # a fresh permutation (and fresh augmentation draws) per epoch, a function of the epoch number alone:
                # the same on every rank, and the same again after a resume
                tgt = loader if hasattr(loader, 'set_epoch') else getattr(loader, 'sampler', None)
                if hasattr(tgt, 'set_epoch'):
                    tgt.set_epoch(self.epoch)
            logging.info(f'Epoch {self.epoch}.')
            train_log, train_batch, train_outputs = self._run_epoch('training')
            logging.info(f'Train log: {train_log}.')
            valid_log, valid_batch, valid_outputs = self._run_epoch('validation')
            logging.info(f'Valid log: {valid_log}.')
            if self.lr_scheduler is not None:
                if isinstance(self.lr_scheduler, torch.optim.lr_scheduler.ReduceLROnPlateau):
                    self.lr_scheduler.step(valid_log['Loss'])
                else:
                    self.lr_scheduler.step()
            # (under torch.distributed the logs are all-reduced in _run_epoch, so the scheduler, the best-checkpoint choice and
            # the early stop below take the same decision on every rank; files are written by rank 0 only)
            if self.logger is not None and dp.rank() == 0:
                self.logger.write(self.epoch, train_log, train_batch, train_outputs, valid_log, valid_batch, valid_outputs)
            if self.monitor is not None:
                path = self.monitor.is_saved(self.epoch)
                if path:
                    self.save(path)
                path = self.monitor.is_best(valid_log)
                if path:
                    self.save(path)
                if self.monitor.is_early_stopped():
                    logging.info('Early stopped.')
                    break
            self.epoch += 1
        if self.logger is not None and dp.rank() == 0:
            self.logger.close()

    def _allocate_data(self, batch):
        if isinstance(batch, dict):
            return {k: self._allocate_data(v) for k, v in batch.items()}
        if isinstance(batch, list):
            return [self._allocate_data(v) for v in batch]
        if isinstance(batch, tuple):
            return tuple(self._allocate_data(v) for v in batch)
        if isinstance(batch, torch.Tensor):
            return batch.to(self.device)
        return batch

    def _init_log(self):
        log = {'Loss': 0}
        for fn in list(self.loss_fns) + list(self.metric_fns):
            log[type(fn).__name__] = 0
        return log

    def save(self, path):
        if dp.rank() != 0:
            return
        torch.save({'net': self.net.state_dict(), 'optimizer': self.optimizer.state_dict(),
                    'lr_scheduler': self.lr_scheduler.state_dict() if self.lr_scheduler else None,
                    'monitor': self.monitor, 'epoch': self.epoch, 'random_state': random.getstate(),
                    'np_random_seeds': self.np_random_seeds}, path)

    def load(self, path):
        ck = torch.load(path, map_location=self.device, weights_only=False)
        self.net.load_state_dict(ck['net'])
        self.optimizer.load_state_dict(ck['optimizer'])
        if ck['lr_scheduler']:
            self.lr_scheduler.load_state_dict(ck['lr_scheduler'])
        self.monitor = ck['monitor']
        self.epoch = ck['epoch'] + 1
        random.setstate(ck['random_state'])
        self.np_random_seeds = ck['np_random_seeds']
