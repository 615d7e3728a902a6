"""DataLoader with per-worker numpy seeding (reference src/data/dataloader.py:51-53) and, under
torch.distributed, a DistributedSampler so that each rank sees its own shard of the samples."""
import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, DistributedSampler


def _seed_worker(worker_id):
    np.random.seed((torch.initial_seed() + worker_id) % (2 ** 32))


class Dataloader(DataLoader):
    def __new__(cls, dataset, batch_size=1, shuffle=False, num_workers=0, collate_fn=None, **kwargs):
        # a dataset that lives in HBM is iterated by the fused gather (one launch per batch), not by worker processes
        if getattr(dataset, 'cache', None) is not None:
            from hipvsr.cine_cache import GpuCineLoader
            return GpuCineLoader(dataset.cache, batch_size=batch_size, shuffle=shuffle, **dataset.loader_kwargs)
        return super().__new__(cls)

    def __init__(self, dataset, batch_size=1, shuffle=False, num_workers=0, collate_fn=None, **kwargs):
        sampler = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            sampler = DistributedSampler(dataset, shuffle=shuffle)
            shuffle = False
        super().__init__(dataset, batch_size=batch_size, shuffle=shuffle, sampler=sampler, num_workers=num_workers,
                         collate_fn=collate_fn, worker_init_fn=_seed_worker, **kwargs)
