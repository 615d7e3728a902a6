"""PSNR (reference src/model/metrics.py:9-36): the parity metric.  Stock torch ops - evaluation only."""
import torch
import torch.nn as nn
import torch.nn.functional as F

__all__ = ['PSNR']


class PSNR(nn.Module):
    def __init__(self, size_average=True, max_value=255):
        super().__init__()
        self.size_average, self.max_value = size_average, max_value

    def forward(self, output, target):
        dims = list(range(1, output.dim()))
        mse = F.mse_loss(output, target, reduction='none').mean(dims)
        psnr = 10 * torch.log10(self.max_value ** 2 / (mse + 1e-10))
        return psnr.mean() if self.size_average else psnr
