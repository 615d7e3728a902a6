"""Loss modules of the RefineNet path (reference src/model/losses.py:5-34).  On a HIP device the trainer routes
L1 / Charbonnier through the fused loss+gradient kernel (hipvsr.autograd.fused_losses); these modules are the
``loss_fn(output, target)`` boundary the reference's config names."""
import torch
import torch.nn as nn

__all__ = ['CharbonnierLoss', 'HuberLoss']


class CharbonnierLoss(nn.Module):
    def __init__(self, epsilon):
        super().__init__()
        self.epsilon = epsilon

    def forward(self, output, target):
        return torch.mean(torch.sqrt((output - target) ** 2 + self.epsilon))


class HuberLoss(nn.Module):
    def __init__(self, delta):
        super().__init__()
        self.delta = delta

    def forward(self, output, target):
        abs_error = torch.abs(output - target)
        quadratic = torch.clamp(abs_error, max=self.delta)
        return torch.mean(0.5 * quadratic ** 2 + self.delta * (abs_error - quadratic))
