"""WelfordRunningStat -- drop-in for rlgym_ppo/util/running_stats.py:15-137 (host-side; it only feeds two scalars
to the hot path: the return std used by GAE and the feature-0 mean/std used to standardise observations)."""
import json
import os

import numpy as np


class WelfordRunningStat(object):
    def __init__(self, shape):
        self.ones = np.ones(shape=shape, dtype=np.float32)
        self.zeros = np.zeros(shape=shape, dtype=np.float32)
        self.running_mean = np.zeros(shape=shape, dtype=np.float32)
        self.running_variance = np.zeros(shape=shape, dtype=np.float32)
        self.count = 0
        self.shape = shape

    def increment(self, samples, num):
        if num > 1:
            for i in range(num):
                self.update(samples[i])
        else:
            self.update(samples)

    def update(self, sample):
        if type(sample) == dict:
            sample = sample["frame"]
        prev = self.count
        self.count = prev + 1
        shape = self.running_mean.shape
        delta = (sample - self.running_mean).reshape(shape)
        delta_n = (delta / self.count).reshape(shape)
        self.running_mean += delta_n
        self.running_variance += delta * delta_n * prev

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
