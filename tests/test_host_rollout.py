"""oracle/host.py::lockstep_rollout (the restatement the device-resident VectorAgentManager is tested against) pinned to
the reference-shaped host path: with ONE one-agent environment the trajectory order is unambiguous, so the restatement
must reproduce BatchedAgentManager + BatchedTrajectory (in-process worker) exactly -- standardisation cadence, raw
first observation, per-trajectory cuts and the forced truncation at the flush included.  CPU only (fake policy)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synthetic_env  # noqa: E402
from oracle import host  # noqa: E402


class _FakePolicy:
    """Deterministic stand-in with the reference's get_action contract (CPU tensors)."""
    def get_action(self, obs, standardize=None):
        obs = np.asarray(obs, np.float32)
        a = (np.abs(obs[:, :7]).sum(1) * 13).astype(np.int64) % 90
        return torch.as_tensor(a), torch.as_tensor(-np.abs(obs[:, 0]).astype(np.float32))


def _act(obs):
    a, lp = _FakePolicy().get_action(obs)
    return a.numpy().astype(np.float32).reshape(-1, 1), lp.numpy()


def test_lockstep_restatement_equals_host_manager_for_one_agent():
    from rlgym_ppo_amd.batched_agents import BatchedAgentManager
    make = lambda: synthetic_env.SyntheticVectorEnv(n_agents=1, seed=4)

    class _AsSingleEnv:  # the worker protocol wants scalar done / truncated
        def __init__(self):
            self.e = make()
            self.observation_space, self.action_space = self.e.observation_space, self.e.action_space
        def reset(self):
            return self._last if getattr(self, "_pending_reset", False) else self.e.reset()
        def step(self, a):
            obs, r, d, t, info = self.e.step(a)
            self._last, self._pending_reset = obs, bool(d[0] or t[0])  # the vector env has already auto-reset
            return obs, [float(r[0])], bool(d[0]), bool(t[0]), info
        def close(self):
            pass

    mgr = BatchedAgentManager(_FakePolicy(), min_inference_size=1, seed=1, standardize_obs=True)
    mgr.init_processes(0, _AsSingleEnv)
    env = make()
    state, reset_obs = None, env.reset()
    for n in (23, 9):
        exp, _, n_col, _ = mgr.collect_timesteps(n)
        ref, state = host.lockstep_rollout(reset_obs, lambda a: env.step(a)[:4], _act, n_col, standardize=True, state=state)
        assert n_col == n
        for got, want, name in zip(exp, ref, ("states", "actions", "log_probs", "rewards", "next_states", "dones", "truncated")):
            got = np.asarray(got, np.float32).reshape(np.asarray(want).shape)
            assert np.array_equal(got, want), name
    mgr.cleanup()
