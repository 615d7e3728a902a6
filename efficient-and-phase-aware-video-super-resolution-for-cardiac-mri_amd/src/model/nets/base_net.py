import torch.nn as nn


class BaseNet(nn.Module):
    """Adds the trainable-parameter summary to repr (reference src/model/nets/base_net.py:11-13)."""

    def __repr__(self):
        n = sum(p.numel() for p in self.parameters() if p.requires_grad)
        return super().__repr__() + f'\nTrainable parameters: {n / 1e6} M\nMemory usage: {(n * 4) / (1 << 20)} MB'
