"""enRainfallLoss (DQWL) — same class / constructor as the reference's models/loss.py:30-57.
One fused HIP pass on the GPU (value and d/dpred together)."""
import torch
import torch.nn as nn

from adnm_hip import ops


class enRainfallLoss(nn.Module):
    def __init__(self, omega_t=0.57, alpha=0.25, gamma=0.1):
        super().__init__()
        self.omega_t, self.alpha, self.gamma = omega_t, alpha, gamma

    def forward(self, pred, target):
        if pred.is_cuda and pred.dtype == torch.float32 and target.dtype == torch.float32 and pred.shape == target.shape:
            return ops.rainloss(pred, target, self.omega_t, self.alpha, self.gamma)  # value + gradient in one HIP pass
        return self.forward_torch(pred, target)

    def forward_torch(self, pred, target):
        """The same expression in device-agnostic torch ops (CPU tensors, odd dtypes)."""
        err = (pred - target).abs()
        over = pred >= target
        w = torch.where(over, 1.0 - self.omega_t, self.omega_t)                # asymmetric L1 (loss.py:40-41)
        heavy = target >= 0.7
        wi = torch.where(heavy, self.alpha * torch.exp(target), torch.zeros_like(target))  # heavy-rain weight (:42-45)
        total = (w * err * (1.0 + wi)).sum()
        if self.gamma != 0.0:                                                   # FN penalty (:49-53)
            fn = torch.where(heavy & ~over, self.gamma * (torch.exp(self.alpha * (target - pred)) - 1.0), torch.zeros_like(pred))
            total = total + fn.sum()
        return total / target.numel()


class RainfallLoss(enRainfallLoss):
    def __init__(self, omega_t=0.57, alpha=0.25):
        super().__init__(omega_t, alpha, gamma=0.0)
