"""RLGymV2GymWrapper -- adapter with the interface of rlgym_ppo/util/rlgym_v2_gym_wrapper.py:5-89: presents an RLGym-v2
environment (dict-of-agents reset/step) as the single gym-style environment the env workers drive.  Out of the accelerated
path ("keep interface-compatible"); gym itself is optional -- without it the two spaces are minimal
stand-ins carrying what the workers read (`.n`, `.shape`, `.seed`), classified by class name like gym's."""
import numpy as np

try:  # pragma: no cover - gym is not installed on the build / GPU boxes
    from gym.spaces import Box, Discrete
except ImportError:
    class Discrete(object):
        def __init__(self, n):
            self.n = int(n)
            self.shape = ()

        def seed(self, seed=None):
            return [seed]

    class Box(object):
        def __init__(self, low, high, shape):
            self.low, self.high, self.shape = low, high, tuple(shape)

        def seed(self, seed=None):
            return [seed]


class RLGymV2GymWrapper(object):
    def __init__(self, rlgym_env):
        self.rlgym_env = rlgym_env
        self.agent_map = {}
        self.obs_buffer = np.zeros(1)
        print("WARNING: CALLING ENV.RESET() ONE EXTRA TIME TO DETERMINE STATE AND ACTION SPACES")
        first_obs = list(rlgym_env.reset().values())
        act_space = next(iter(rlgym_env.action_spaces.values()))[1]
        obs_space = next(iter(rlgym_env.observation_spaces.values()))[1]
        self.is_discrete = type(act_space) == int
        self.action_space = Discrete(n=act_space) if self.is_discrete else None
        if type(obs_space) == int and obs_space > 0:
            self.observation_space = Box(low=-np.inf, high=np.inf, shape=(obs_space,))
        elif first_obs:
            self.observation_space = Box(low=-np.inf, high=np.inf, shape=np.shape(first_obs[0]))
        else:
            self.observation_space = None

    def reset(self):
        obs_dict = self.rlgym_env.reset()
        self.agent_map = dict(enumerate(obs_dict.keys()))
        self.obs_buffer = np.asarray(list(obs_dict.values()))
        return self.obs_buffer

    def step(self, actions):
        if self.is_discrete:
            actions = actions.astype(np.int32)
        action_dict = {self.agent_map[i]: actions[i] for i in range(len(actions))}
        obs_dict, reward_dict, terminated_dict, truncated_dict = self.rlgym_env.step(action_dict)
        rews, done, truncated = [], False, False
        for i, (agent_id, agent_obs) in enumerate(obs_dict.items()):
            self.obs_buffer[i] = agent_obs
            rews.append(reward_dict[agent_id])
            done = done or terminated_dict[agent_id]
            truncated = truncated or truncated_dict[agent_id]
        return self.obs_buffer, rews, done, truncated, {"state": self.rlgym_env.state}

    def render(self):
        self.rlgym_env.render()

    def seed(self, seed):
        pass

    def close(self):
        self.rlgym_env.close()
