"""oracle/host.py -- TEST INFRASTRUCTURE (see oracle/__init__.py).

numpy restatement of the two host-side data structures that feed the hot path:

    fifo_append / shuffled_batches   ExperienceBuffer._cat and get_all_batches_shuffled
                                     (reference: rlgym_ppo/ppo/experience_buffer.py:18-37, 89-102)
    Welford                          WelfordRunningStat (reference: rlgym_ppo/util/running_stats.py:15-137)

The shuffle is numpy's *legacy* `RandomState(seed).permutation(n)` (MT19937 + masked-rejection
Fisher-Yates); numpy is a third-party dependency of the reference (requirements.txt:7, `numpy<2.0`) and is
present on every box, so it is used directly as the index oracle for the product's own C implementation.
"""
import numpy as np


def fifo_append(old, new, size):
    """Keep the newest `size` rows of old ++ new."""
    new = np.asarray(new, np.float32)
    if old is None or len(old) == 0:
        old = new[:0]
    if len(new) >= size:
        return new[len(new) - size:].copy()
    keep = min(len(old), size - len(new))
    return np.concatenate([old[len(old) - keep:], new], 0)


def shuffled_batches(rng, total, batch_size):
    """Index arrays of one epoch; the tail that does not fill a batch is dropped (quirk Q7)."""
    idx = rng.permutation(total)
    return [idx[s:s + batch_size] for s in range(0, total - batch_size + 1, batch_size)]


class Welford:
    def __init__(self, shape):
        self.shape = shape
        self.mean_ = np.zeros(shape, np.float32)
        self.m2 = np.zeros(shape, np.float32)
        self.count = 0

    def update(self, sample):
        n0 = self.count
        self.count += 1
        delta = (sample - self.mean_).reshape(self.mean_.shape)
        delta_n = (delta / self.count).reshape(self.mean_.shape)
        # in-place adds keep the float32 state dtype exactly like `+=` on a float32 ndarray does
        np.add(self.mean_, delta_n, out=self.mean_, casting="same_kind")
        np.add(self.m2, delta * delta_n * n0, out=self.m2, casting="same_kind")

    def increment(self, samples, num):
        if num > 1:
            for i in range(num):
                self.update(samples[i])
        else:
            self.update(samples)

    @property
    def mean(self):
        return np.zeros(self.shape, np.float32) if self.count < 2 else self.mean_

    @property
    def std(self):
        if self.count < 2:
            return np.ones(self.shape, np.float32)
        var = self.m2 / (self.count - 1)
        return np.sqrt(np.where(var == 0, 1.0, var))
