"""``python -m src.main <config.yaml>``: the reference's CLI and YAML surface for the RefineNet training path
(reference src/main.py:19-190): every section ``{name, kwargs}`` is instantiated by name from the matching
namespace; losses are looked up in torch.nn first; ``main.random_seed`` may be a string.  ``python-box`` is not
in this image, so the YAML is wrapped in a small attribute dict.  ``--test`` runs the whole-cycle predictor.  Under ``torchrun`` (WORLD_SIZE > 1) one process
per GPU is used and gradients are all-reduced (hipvsr.dp)."""
import argparse
import logging
import os
import random
from pathlib import Path

import torch
import yaml

import src


class Cfg(dict):
    """dict with attribute access, recursively (stand-in for box.Box)."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = self._wrap(v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict):
            return cls(v)
        if isinstance(v, list):
            return [cls._wrap(x) for x in v]
        return v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def to_dict(self):
        def un(v):
            if isinstance(v, dict):
                return {k: un(x) for k, x in v.items()}
            if isinstance(v, list):
                return [un(x) for x in v]
            return str(v) if isinstance(v, Path) else v
        return un(self)


def _get_instance(module, config, *args):
    cls = getattr(module, config.name)
    kwargs = config.get('kwargs')
    return cls(*args, **kwargs) if kwargs else cls(*args)


def _get_optimizer(config, params, device):
    """``torch.optim.<name>(params, **kwargs)`` as the reference does (src/main.py:76); plain Adam on a HIP device is
    served by the flat single-launch implementation (same constructor, same state_dict format)."""
    kwargs = dict(config.get('kwargs') or {})
    plain = not any(kwargs.get(k) for k in ('amsgrad', 'maximize', 'capturable', 'differentiable'))
    if config.name == 'Adam' and device.type == 'cuda' and plain:
        from hipvsr.step_tail import FlatAdam
        return FlatAdam(params, **kwargs)
    return _get_instance(torch.optim, config, params)


def _losses_metrics(config):
    loss_fns, loss_weights = [], []
    torch_losses = [n for n in dir(torch.nn) if 'Loss' in n]
    for cl in config.losses:
        loss_fns.append(_get_instance(torch.nn if cl.name in torch_losses else src.model.losses, cl))
        loss_weights.append(cl.weight)
    return loss_fns, loss_weights


def _test(config):
    """The --test branch (reference src/main.py:113-160): test split, batch-1 loader, net, losses, ALL configured metrics
    (Cardiac* included), the predictor named in the ``predictor`` section, checkpoint from ``main.loaded_path``.
    Whole-cycle inference does not shard: under torchrun every rank would run a replica, so only one process is used."""
    if 'cuda' in config.predictor.kwargs.device and not torch.cuda.is_available():
        raise ValueError("The cuda is not available. Please set the device in the predictor section to 'cpu'.")
    device = torch.device(config.predictor.kwargs.device)
    if device.type == 'cuda':
        torch.cuda.set_device(device)
    config.dataset.kwargs.update(data_dir=config.dataset.kwargs.get('data_dir'), type='test', device=device)
    test_dataset = _get_instance(src.data.datasets, config.dataset)
    test_loader = _get_instance(src.data.dataloader, config.dataloader, test_dataset)
    net = _get_instance(src.model.nets, config.net)
    loss_fns, loss_weights = _losses_metrics(config)
    metric_fns = [_get_instance(src.model.metrics, cm) for cm in config.metrics]
    config.predictor.kwargs.update(device=device, test_dataloader=test_loader, net=net, loss_fns=loss_fns,
                                   loss_weights=loss_weights, metric_fns=metric_fns)
    predictor = _get_instance(src.runner.predictors, config.predictor)
    if config.net.name != 'Bicubic':
        logging.info(f'Load the previous checkpoint from "{config.main.loaded_path}".')
        predictor.load(Path(config.main.loaded_path))
    logging.info('Start testing.')
    log = predictor.predict()
    logging.info('End testing.')
    return log


def main(args):
    with open(args.config_path) as f:
        config = Cfg(yaml.safe_load(f))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    saved_dir = Path(config.main.saved_dir)
    if rank == 0:
        saved_dir.mkdir(parents=True, exist_ok=True)
        with open(saved_dir / 'config.yaml', 'w+') as f:
            yaml.dump(config.to_dict(), f, default_flow_style=False)
    if args.test:
        # whole-cycle inference does not shard (replicas only): under torchrun one process does the work
        return _test(config) if rank == 0 else None

    if world > 1:
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
            from hipvsr.hip_ops import touch_side_streams
            touch_side_streams(f'cuda:{local_rank}')              # before RCCL makes its stream: hardware-queue pairing as in the single-GPU run
            torch.distributed.init_process_group('nccl', device_id=torch.device(f'cuda:{local_rank}'))
        else:
            torch.distributed.init_process_group('gloo')

    random.seed(config.main.random_seed)
    torch.manual_seed(random.getstate()[1][1])
    if 'cuda' in config.trainer.kwargs.device and not torch.cuda.is_available():
        raise ValueError("The cuda is not available. Please set the device in the trainer section to 'cpu'.")
    dev = config.trainer.kwargs.device
    if world > 1 and 'cuda' in dev:
        dev = f'cuda:{local_rank}'
    device = torch.device(dev)
    if device.type == 'cuda':
        torch.cuda.set_device(device)

    data_dir = config.dataset.kwargs.get('data_dir')
    config.dataset.kwargs.update(data_dir=data_dir, type='train', device=device, loader_seed=str(config.main.random_seed))
    train_dataset = _get_instance(src.data.datasets, config.dataset)
    config.dataset.kwargs.update(type='valid')
    valid_dataset = _get_instance(src.data.datasets, config.dataset)
    cls = getattr(src.data.datasets, config.dataset.name)
    tb, vb = config.dataloader.kwargs.pop('train_batch_size'), config.dataloader.kwargs.pop('valid_batch_size')
    config.dataloader.kwargs.update(collate_fn=getattr(cls, 'collate_fn', None), batch_size=tb)
    train_loader = _get_instance(src.data.dataloader, config.dataloader, train_dataset)
    config.dataloader.kwargs.update(batch_size=vb)
    valid_loader = _get_instance(src.data.dataloader, config.dataloader, valid_dataset)

    net = _get_instance(src.model.nets, config.net)
    loss_fns, loss_weights = _losses_metrics(config)
    metric_fns = [_get_instance(src.model.metrics, cm) for cm in config.metrics]      # an unknown name raises, as in the reference
    optimizer = _get_optimizer(config.optimizer, net.parameters(), device)
    lr_scheduler = _get_instance(torch.optim.lr_scheduler, config.lr_scheduler, optimizer) if config.get('lr_scheduler') else None
    config.logger.kwargs.update(log_dir=saved_dir / 'log', net=net)
    logger = _get_instance(src.callbacks.loggers, config.logger) if rank == 0 else None
    config.monitor.kwargs.update(checkpoints_dir=saved_dir / 'checkpoints')
    monitor = _get_instance(src.callbacks.monitor, config.monitor)
    config.trainer.kwargs.update(device=device, train_dataloader=train_loader, valid_dataloader=valid_loader, net=net,
                                 loss_fns=loss_fns, loss_weights=loss_weights, metric_fns=metric_fns, optimizer=optimizer,
                                 lr_scheduler=lr_scheduler, logger=logger, monitor=monitor)
    trainer = _get_instance(src.runner.trainers, config.trainer)
    loaded_path = config.main.get('loaded_path')
    if loaded_path:
        trainer.load(Path(loaded_path))
    trainer.train()


def _parse_args():
    p = argparse.ArgumentParser(description='The script for the training and the testing.')
    p.add_argument('config_path', type=Path, help='The path of the config file.')
    p.add_argument('--test', action='store_true', help='Perform the testing if specified; otherwise perform the training.')
    return p.parse_args()


if __name__ == '__main__':
    logging.basicConfig(format='%(asctime)s | %(levelname)s | %(message)s', level=logging.INFO, datefmt='%Y-%m-%d %H:%M:%S')
    main(_parse_args())
