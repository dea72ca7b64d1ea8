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
        # the product path is the fused HIP pass (value + gradient); like every other adnm_hip op it has no CPU / other-dtype fallback
        if not (pred.is_cuda and target.is_cuda):
            raise RuntimeError("enRainfallLoss: adnm_hip kernels run on the GPU only (there is no CPU fallback); got CPU tensors "
                               "(forward_torch() states the same expression in torch ops for checks)")
        if pred.dtype != torch.float32 or target.dtype != torch.float32 or pred.shape != target.shape:
            raise RuntimeError(f"enRainfallLoss: needs fp32 pred / target of one shape, got {pred.dtype} {tuple(pred.shape)}, "
                               f"{target.dtype} {tuple(target.shape)}")
        return ops.rainloss(pred, target, self.omega_t, self.alpha, self.gamma)

    def forward_torch(self, pred, target):
        """The same expression in device-agnostic torch ops: a checker for the tests (fp64 on the CPU), never called by forward()."""
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


class Weighted_mse_mae(nn.Module):
    """The baselines' loss (loss.py:73-98 of the reference; ConvLSTM recipe train_untils.py:57-61) — BASELINE config 1's
    plumbing case.  Plain torch: it is not on the hot path.  input / target: (B, S, C, H, W)."""

    def __init__(self, mse_weight=1.0, mae_weight=1.0, NORMAL_LOSS_GLOBAL_SCALE=0.00005, LAMBDA=None, thresholds=[]):
        super().__init__()
        self.NORMAL_LOSS_GLOBAL_SCALE, self.mse_weight, self.mae_weight = NORMAL_LOSS_GLOBAL_SCALE, mse_weight, mae_weight
        self._lambda, self.thresholds = LAMBDA, thresholds

    def forward(self, input, target):
        steps = (1, 1, 2, 5, 10, 30)   # balancing weights: +1, +3, +5, +20 as the target crosses successive thresholds
        w = torch.full_like(input, float(steps[0]))
        for i, th in enumerate(self.thresholds):
            w = w + (steps[i + 1] - steps[i]) * (target >= th).to(input.dtype)
        d = input - target
        mse = (w * d * d).sum((2, 3, 4)).t()    # (S, B)
        mae = (w * d.abs()).sum((2, 3, 4)).t()
        if self._lambda is not None:            # later frames weigh more
            ramp = (1.0 + self._lambda * torch.arange(mse.shape[0], device=mse.device, dtype=mse.dtype)).unsqueeze(1)
            mse, mae = mse * ramp, mae * ramp
        return self.NORMAL_LOSS_GLOBAL_SCALE * (self.mse_weight * mse.mean() + self.mae_weight * mae.mean())
