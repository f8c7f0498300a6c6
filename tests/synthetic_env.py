"""A gym-free synthetic environment speaking the interface the env workers expect (reset/step/observation_space/
action_space), so the whole Learner loop can run without RocketSim or gym.  Deterministic given its seed."""
import numpy as np


class _Space:
    def __init__(self, shape=None, n=None):
        self.shape, self._n = shape, n
        if n is not None:
            self.n = n

    def seed(self, s):
        pass


class Discrete(_Space):
    pass


class MultiDiscrete(_Space):
    pass


class Box(_Space):
    pass


class SyntheticEnv:
    def __init__(self, obs_dim=107, n_actions=90, n_agents=2, ep_len=17, seed=0, kind="discrete", new_gym_api=True):
        self.obs_dim, self.n_agents, self.ep_len, self.kind = obs_dim, n_agents, ep_len, kind
        self.rs = np.random.RandomState(seed)
        self.observation_space = _Space(shape=(obs_dim,))
        if kind == "discrete":
            self.action_space = Discrete(n=n_actions)
        elif kind == "multidiscrete":
            self.action_space = MultiDiscrete(shape=(8,))
        else:
            self.action_space = Box(shape=(n_actions,))
        self.new_gym_api = new_gym_api
        self.t = 0

    def _obs(self):
        return (self.rs.randn(self.n_agents, self.obs_dim) * 2 + 0.5).astype(np.float32)

    def reset(self):
        self.t = 0
        return self._obs()

    def step(self, actions):
        actions = np.asarray(actions)
        assert actions.shape[0] == self.n_agents
        self.t += 1
        rew = [float(np.tanh(np.sum(actions[i]) * 0.01) + self.rs.randn() * 0.1) for i in range(self.n_agents)]
        done = self.t >= self.ep_len
        truncated = (not done) and (self.t % 11 == 0)
        if self.new_gym_api:
            return self._obs(), rew, done, truncated, {"state": None}
        return self._obs(), rew, done, {"state": None}

    def close(self):
        pass


def make_discrete_env():
    return SyntheticEnv(kind="discrete")


def make_continuous_env():
    return SyntheticEnv(obs_dim=231, n_actions=8, n_agents=3, kind="continuous")


def make_multidiscrete_env():
    return SyntheticEnv(kind="multidiscrete", new_gym_api=False)


class SyntheticVectorEnv:
    """n agents stepping in lockstep with auto-reset: observations are drawn independently of the actions (so two
    implementations of the policy that disagree on a near-tie still see the same observation stream), rewards depend on
    the actions, episode lengths differ per agent."""

    def __init__(self, obs_dim=107, n_actions=90, n_agents=16, seed=0, kind="discrete"):
        self.obs_dim, self.n_agents, self.kind = obs_dim, n_agents, kind
        self.rs = np.random.RandomState(seed)
        self.observation_space = _Space(shape=(obs_dim,))
        self.action_space = Discrete(n=n_actions) if kind == "discrete" else Box(shape=(n_actions,))
        self.ep_len = 5 + (np.arange(n_agents) * 7) % 13
        self.t = np.zeros(n_agents, np.int64)

    def _obs(self):
        return (self.rs.randn(self.n_agents, self.obs_dim) * 2 + 0.5).astype(np.float32)

    def reset(self):
        self.t[:] = 0
        return self._obs()

    def step(self, actions):
        actions = np.asarray(actions, np.float32).reshape(self.n_agents, -1)
        self.t += 1
        rew = (np.tanh(actions.sum(1) * 0.01) + self.rs.randn(self.n_agents) * 0.1).astype(np.float32)
        done = self.t >= self.ep_len
        trunc = (~done) & (self.t % 4 == 0) & (np.arange(self.n_agents) % 3 == 0)
        self.t[done | trunc] = 0
        return self._obs(), rew, done.astype(np.float32), trunc.astype(np.float32), {"state": None}

    def close(self):
        pass


def make_vector_env():
    return SyntheticVectorEnv()


class SyntheticSingleEnv:
    """ONE agent, rank-1 observations and a scalar reward (the shape a plain gym environment has): exercises the rank-1
    branches of the worker <-> learner wire format; `info["state"]` carries the step count for a metrics function."""

    def __init__(self, obs_dim=11, n_actions=5, ep_len=6, seed=0):
        self.obs_dim, self.ep_len = obs_dim, ep_len
        self.rs = np.random.RandomState(seed)
        self.observation_space = _Space(shape=(obs_dim,))
        self.action_space = Discrete(n=n_actions)
        self.t = 0

    def _obs(self):
        return (self.rs.randn(self.obs_dim) * 2 + 0.5).astype(np.float32)

    def reset(self):
        self.t = 0
        return self._obs()

    def step(self, actions):
        self.t += 1
        rew = float(np.tanh(float(np.sum(actions)) * 0.1) + self.rs.randn() * 0.1)
        done = self.t >= self.ep_len
        truncated = (not done) and self.t == 4
        return self._obs(), rew, done, truncated, {"state": self.t}

    def close(self):
        pass


def make_single_env():
    return SyntheticSingleEnv()


def make_wire_env():
    return SyntheticEnv(obs_dim=13, n_actions=7, n_agents=2, ep_len=5, seed=2)


def step_count_metrics(state):
    """A metrics function (what Learner passes as collect_metrics_fn): float32 vector of length 3."""
    return np.asarray([state, 2.0 * state, -1.0], dtype=np.float32)


class SyntheticVaryingEnv(SyntheticEnv):
    """A match whose number of agents changes at every reset (2 -> 3 -> 1 -> 2 ...): the collector must flush the environment's
    trajectory and pad the last step's next states (batched_agent_manager.py: `nxt.shape[0] != prev_n`)."""

    def __init__(self):
        super().__init__(obs_dim=13, n_actions=7, n_agents=2, ep_len=4, seed=3)
        self._cycle, self._k = (2, 3, 1), 0

    def reset(self):
        self.n_agents = self._cycle[self._k % 3]
        self._k += 1
        return super().reset()

    def step(self, actions):
        obs, rew, done, truncated, info = super().step(actions)
        info = {"state": self.t}
        return obs, rew, done, truncated, info


def make_varying_env():
    return SyntheticVaryingEnv()


def make_continuous_wire_env():
    return SyntheticEnv(obs_dim=13, n_actions=3, n_agents=2, ep_len=6, seed=4, kind="continuous")
