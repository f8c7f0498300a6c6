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


def test_trajectory_assembler_semantics():
    """BatchedTrajectory as rlgym_ppo/batched_agents/batched_trajectory.py:23-104 specifies it: a timestep is banked only when all
    seven fields are there; a scalar reward becomes a one-element list; `truncated` keeps its value across the reset of the other
    six; update() is True exactly when the banked step was terminal; get_all() splits the match's steps into one trajectory per
    agent (seven parallel lists) and empties the assembler."""
    from rlgym_ppo_amd.batched_agents.batched_trajectory import BatchedTrajectory
    t = BatchedTrajectory()
    assert t.update() is False and t.get_all() == []
    s0, s1 = np.arange(6, dtype=np.float32).reshape(2, 3), np.arange(6, 12, dtype=np.float32).reshape(2, 3)
    t.state, t.action, t.log_prob = s0, np.array([[1.0], [2.0]], np.float32), np.array([-0.5, -0.25], np.float32)
    assert t.update() is False                      # the environment's half is missing
    t.reward, t.next_state, t.done, t.truncated = [0.5, -1.0], s1, 0.0, 1.0
    assert t.update() is False and len(t.complete_timesteps) == 1
    assert t.state is None and t.reward is None and t.done is None and t.truncated == 1.0
    t.state, t.action, t.log_prob = s1, np.array([[3.0], [4.0]], np.float32), np.array([-0.75, -1.5], np.float32)
    t.reward, t.next_state, t.done = [1.0, 2.0], s0, 1.0
    assert t.update() is True                       # terminal step; truncated still holds 1.0 from the step before
    out = t.get_all()
    assert len(out) == 2 and t.complete_timesteps == [] and t.get_all() == []
    for i, cols in enumerate(out):
        states, actions, log_probs, rewards, next_states, dones, truncs = cols
        assert np.array_equal(states[0], s0[i]) and np.array_equal(states[1], s1[i]) and np.array_equal(next_states[1], s0[i])
        assert [float(a[0]) for a in actions] == [1.0 + i, 3.0 + i] and [float(x) for x in log_probs] == [[-0.5, -0.75], [-0.25, -1.5]][i]
        assert rewards == [[0.5, 1.0], [-1.0, 2.0]][i] and dones == [0.0, 1.0] and truncs == [1.0, 1.0]
    one = BatchedTrajectory()                       # a single agent: scalar reward
    one.state, one.action, one.log_prob, one.reward, one.next_state, one.done, one.truncated = s0[:1], np.zeros((1, 1)), np.zeros(1), 0.25, s1[:1], 0.0, 0.0
    assert one.update() is False and one.complete_timesteps[0][3] == [0.25]
