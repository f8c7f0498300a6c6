"""Host-side pieces that need no GPU: the RLGym-v2 adapter (rlgym_ppo/util/__init__.py:1-4 exports it, so a user script's
`from rlgym_ppo.util import RLGymV2GymWrapper` must keep working after the import swap) and the ExperienceBuffer's ring
against the oracle's restatement of the reference FIFO (experience_buffer.py:18-37)."""
import numpy as np
import torch

from oracle import host


class _V2Env:
    """Minimal RLGym-v2 shaped environment: dict observations / rewards / flags keyed by agent id."""

    def __init__(self):
        self.agents = ["blue-0", "orange-0", "orange-1"]
        self.action_spaces = {a: ("discrete", 90) for a in self.agents}
        self.observation_spaces = {a: ("real", 7) for a in self.agents}
        self.state, self.t, self.last_actions = "S0", 0, None

    def _obs(self):
        return {a: np.full(7, self.t + i, np.float32) for i, a in enumerate(self.agents)}

    def reset(self):
        self.t = 0
        return self._obs()

    def step(self, actions):
        self.t += 1
        self.last_actions = actions
        self.state = f"S{self.t}"
        done = self.t >= 3
        return (self._obs(), {a: float(i) for i, a in enumerate(self.agents)}, {a: done for a in self.agents},
                {a: False for a in self.agents})

    def close(self):
        pass


def test_rlgym_v2_gym_wrapper_interface():
    from rlgym_ppo_amd.batched_agents.batched_agent import describe_action_space
    from rlgym_ppo_amd.util import KBHit, MetricsLogger, RLGymV2GymWrapper, WelfordRunningStat  # the reference's four exports
    assert all(x is not None for x in (KBHit, MetricsLogger, WelfordRunningStat))
    env = RLGymV2GymWrapper(_V2Env())
    assert env.is_discrete and env.action_space.n == 90 and env.observation_space.shape == (7,)
    assert describe_action_space(env.action_space) == (90.0, 0.0)
    obs = env.reset()
    assert obs.shape == (3, 7) and env.agent_map == {0: "blue-0", 1: "orange-0", 2: "orange-1"}
    obs, rews, done, truncated, info = env.step(np.array([5.0, 6.0, 7.0], np.float32))
    assert env.rlgym_env.last_actions == {"blue-0": 5, "orange-0": 6, "orange-1": 7}
    assert obs.shape == (3, 7) and obs[2, 0] == 3 and rews == [0.0, 1.0, 2.0] and not done and not truncated
    assert info == {"state": "S1"}
    env.step(np.zeros(3, np.float32))
    assert env.step(np.zeros(3, np.float32))[2] is True
    env.close()


def test_ring_buffer_keeps_the_reference_fifo_order():
    """The ring (rotating base index, geometric growth up to max_size, in-place overwrite of the oldest rows) holds exactly the
    rows the reference's _cat would hold, in its order, through every one of its four cases; the (base, capacity) map the
    kernels apply to a permutation's logical rows lands on the same rows."""
    import rlgym_ppo_amd.ppo.experience_buffer as eb

    class HostRing(eb.ExperienceBuffer):  # same code, CPU storage: the class itself insists on a GPU
        def __init__(self, max_size):
            self._dev, self.max_size = torch.device("cpu"), max_size
            self._store = {k: None for k in eb._FIELDS}
            self._cap = self._base = self._count = 0
            self._d = 2

        def _pad_states(self, x):
            return torch.as_tensor(np.asarray(x, np.float32)).reshape(len(x), -1)

    rs = np.random.RandomState(0)
    for size in (10, 7, 100, 33, 1):
        buf, ref, nxt, caps = HostRing(size), None, 0, set()
        for it in range(60):
            c = int(rs.choice([1, 2, 3, 5, 8, max(size - 1, 1), size, size + 3, 2 * size + 1]))
            ar = np.arange(nxt, nxt + c, dtype=np.float32)
            nxt += c
            buf.submit_experience(ar[:, None].repeat(2, 1), ar, ar, ar, ar[:, None].repeat(2, 1), ar, ar, ar, ar)
            ref = host.fifo_append(ref, ar, size)
            assert len(buf) == len(ref) and np.array_equal(buf.rewards.numpy(), ref), (size, it)
            assert np.array_equal(buf.states.numpy()[:, 1], ref) and np.array_equal(buf.next_states.numpy()[:, 0], ref)
            store, base, cap = buf.ring()
            caps.add(cap)
            assert cap <= size and 0 <= base < cap
            phys = (np.arange(len(buf)) + base) % cap
            assert np.array_equal(store["advantages"].numpy()[phys], ref)
        assert max(caps) == size  # no allocation beyond max_size, and once there the storage is never re-allocated
    buf = HostRing(8)
    try:
        buf.submit_experience(np.zeros((3, 2)), np.zeros(3), np.zeros(2), np.zeros(3), np.zeros((3, 2)), np.zeros(3), np.zeros(3),
                              np.zeros(3), np.zeros(3))
        raise AssertionError("ragged fields must be rejected")
    except ValueError:
        pass
