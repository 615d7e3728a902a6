"""Checkpoint / early-stop policy (reference src/callbacks/monitor.py:26-63): periodic ``model_{epoch}.pth`` every
``saved_freq`` epochs, ``model_best.pth`` when the target improves, early stop after ``early_stop`` stale epochs."""
import math
from pathlib import Path


class Monitor:
    def __init__(self, checkpoints_dir, mode, target, saved_freq, early_stop=0):
        self.checkpoints_dir = Path(checkpoints_dir)
        if mode not in ('min', 'max'):
            raise ValueError(f"The mode should be 'min' or 'max'. Got {mode}.")
        self.mode, self.target, self.saved_freq = mode, target, saved_freq
        self.early_stop = math.inf if early_stop == 0 else early_stop
        self.best = math.inf if mode == 'min' else -math.inf
        self.not_improved_count = 0
        self.checkpoints_dir.mkdir(parents=True, exist_ok=True)

    def is_saved(self, epoch):
        return self.checkpoints_dir / f'model_{epoch}.pth' if epoch % self.saved_freq == 0 else None

    def is_best(self, valid_log):
        score = valid_log[self.target]
        if (self.mode == 'min' and score < self.best) or (self.mode == 'max' and score > self.best):
            self.best, self.not_improved_count = score, 0
            return self.checkpoints_dir / 'model_best.pth'
        self.not_improved_count += 1
        return None

    def is_early_stopped(self):
        return self.not_improved_count == self.early_stop
