"""The per-process environment of bench.py's `process_collect` leg: the reference's default shape (a 1v1 match: two agents, 107-float
observations, 90 discrete actions) with pre-drawn observations, so that the leg measures the collector -- worker processes, the wire
format, get_action on what the ready workers hand in -- and not numpy's randn.  Imported by the worker processes (forkserver / spawn)."""
import numpy as np

OBS, ACT, AGENTS, EP_LEN = 107, 90, 2, 300


class _Space:
    def __init__(self, shape=None, n=None):
        self.shape = shape
        if n is not None:
            self.n = n

    def seed(self, s):
        pass


class BenchProcessEnv:
    def __init__(self, seed=0):
        self.observation_space = _Space(shape=(OBS,))
        self.action_space = _Space(n=ACT)
        self._pool = (np.random.RandomState(seed).randn(64, AGENTS, OBS) * 2 + 0.5).astype(np.float32)
        self._i = 0
        self.t = 0

    def _obs(self):
        self._i += 1
        return self._pool[self._i % 64]

    def reset(self):
        self.t = 0
        return self._obs()

    def step(self, actions):
        self.t += 1
        done = self.t >= EP_LEN
        return self._obs(), [0.1] * AGENTS, done, False, {"state": None}

    def close(self):
        pass


def make_env():
    return BenchProcessEnv()
