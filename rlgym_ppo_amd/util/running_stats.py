"""WelfordRunningStat -- drop-in for rlgym_ppo/util/running_stats.py:15-137 (host-side; it only feeds two scalars
to the hot path: the return std used by GAE and the feature-0 mean/std used to standardise observations)."""
import json
import os

import numpy as np


class WelfordRunningStat(object):
    """Attribute names (`running_mean`, `running_variance`, `count`, `shape`, `ones`, `zeros`), the JSON keys and the float32 operation
    ORDER of `update` / `std` are the contract (checkpoints interchange with the reference's; the return std GAE divides by is
    bit-identical: tests/test_abi_and_layout.py::test_welford_matches_reference, fixture G7); everything else is this file's own."""

    def __init__(self, shape):
        self.shape, self.count = shape, 0
        # float32 state and float32 constants: `mean` / `std` hand these very arrays out while fewer than two samples have been seen
        self.zeros, self.ones = (np.full(shape, fill, dtype=np.float32) for fill in (0.0, 1.0))
        self.running_mean, self.running_variance = self.zeros.copy(), self.zeros.copy()

    def increment(self, samples, num):
        # num > 1: samples[0 .. num) one by one; otherwise `samples` IS the one sample (running_stats.py:28-33)
        for sample in ([samples[i] for i in range(num)] if num > 1 else [samples]):
            self.update(sample)

    def update(self, sample):
        if type(sample) is dict:     # a frame-stacked observation: its newest frame (running_stats.py:36-37)
            sample = sample["frame"]
        seen = self.count            # samples before this one
        self.count = seen + 1
        like = self.running_mean.shape
        # Welford's recurrence in the reference's operation order (float32 arrays: every step rounds where the reference's does)
        delta = (sample - self.running_mean).reshape(like)
        delta_n = (delta / self.count).reshape(like)
        self.running_mean += delta_n
        self.running_variance += delta * delta_n * seen

    def reset(self):
        self.__init__(self.shape)

    @property
    def mean(self):
        return self.zeros if self.count < 2 else self.running_mean

    @property
    def std(self):
        if self.count < 2:
            return self.ones
        var = self.running_variance / (self.count - 1)
        return np.sqrt(np.where(var == 0, 1.0, var))

    def increment_from_serialized_other(self, serialized_other):
        n = int(np.prod(self.shape))
        other_mean = np.asarray(serialized_other[:n], dtype=np.float32).reshape(self.running_mean.shape)
        other_var = np.asarray(serialized_other[n:-1], dtype=np.float32).reshape(self.running_variance.shape)
        other_count = serialized_other[-1]
        if other_count == 0:
            return
        count = self.count + other_count
        d = other_mean - self.running_mean
        self.running_variance = self.running_variance + other_var + d * d * self.count * other_count / count
        self.running_mean = (self.count * self.running_mean + other_count * other_mean) / count
        self.count = count

    def serialize(self):
        return self.running_mean.ravel().tolist() + self.running_variance.ravel().tolist() + [self.count]

    def deserialize(self, other):
        self.reset()
        n = int(np.prod(self.shape))
        self.running_mean = np.reshape(other[:n], self.shape)
        self.running_variance = np.reshape(other[n:-1], self.shape)
        self.count = other[-1]

    def to_json(self):
        return {"mean": self.running_mean.ravel().tolist(), "var": self.running_variance.ravel().tolist(),
                "shape": np.shape(self.running_mean), "count": self.count}

    def from_json(self, other_json):
        shape = other_json["shape"]
        self.count = other_json["count"]
        self.running_mean = np.asarray(other_json["mean"]).reshape(shape)
        self.running_variance = np.asarray(other_json["var"]).reshape(shape)
        print(f"LOADED RUNNING STATS FROM JSON | Mean: {self.running_mean} | Variance: {self.running_variance} | Count: {self.count}")

    def save(self, directory):
        with open(os.path.join(directory, "RUNNING_STATS.json"), "w") as f:
            json.dump(obj=self.to_json(), fp=f, indent=4)

    def load(self, directory):
        with open(os.path.join(directory, "RUNNING_STATS.json"), "r") as f:
            self.from_json(dict(json.load(f)))
