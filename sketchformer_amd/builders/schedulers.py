"""builders/schedulers.py of the reference (:13-46), evaluated in float32 like TF does."""
import math

import numpy as np


class WarmupDecay(object):
    """lrate = d**-0.5 * min(step**-0.5, step * warmup_steps**-1.5); step 0 -> 0 (Keras passes ``iterations``
    before the increment, so the first update has lr = 0)."""

    def __init__(self, d_model, warmup_steps=4000):
        self.d_model = np.float32(d_model)
        self.warmup_steps = warmup_steps

    def __call__(self, step):
        step = np.float32(step)
        with np.errstate(divide="ignore"):
            arg1 = np.float32(1.0) / np.sqrt(step)
        arg2 = step * np.float32(self.warmup_steps ** -1.5)
        return np.float32(np.float32(1.0) / np.sqrt(self.d_model) * np.minimum(arg1, arg2))


class StepDecay(object):
    def __init__(self, init_lr, decay_rate=0.1, decay_steps=50000, min_lr_ratio=1e-2):
        self.init_lr, self.decay_rate, self.decay_steps, self.min_lr_ratio = init_lr, decay_rate, decay_steps, min_lr_ratio

    def __call__(self, step):
        return max(self.init_lr * self.decay_rate ** math.floor(step / self.decay_steps), self.init_lr * self.min_lr_ratio)
