"""Linear warm-up + cosine annealing, stepped once per epoch (reference: models/lr_scheduler.py:9-115; the
chainable and closed forms there yield the same sequence -- this is the closed form)."""
import math

from torch.optim.lr_scheduler import LRScheduler


class LinearWarmupCosineAnnealingLR(LRScheduler):
    def __init__(self, optimizer, warmup_epochs, max_epochs, warmup_start_lr=0.0, eta_min=0.0, last_epoch=-1):
        self.warmup_epochs, self.max_epochs = warmup_epochs, max_epochs
        self.warmup_start_lr, self.eta_min = warmup_start_lr, eta_min
        super().__init__(optimizer, last_epoch)

    def _at(self, base_lr, e):
        if e < self.warmup_epochs:
            return self.warmup_start_lr + e * (base_lr - self.warmup_start_lr) / max(1, self.warmup_epochs - 1)
        span = self.max_epochs - self.warmup_epochs
        return self.eta_min + 0.5 * (base_lr - self.eta_min) * (1.0 + math.cos(math.pi * (e - self.warmup_epochs) / span))

    def get_lr(self):
        return [self._at(b, self.last_epoch) for b in self.base_lrs]
