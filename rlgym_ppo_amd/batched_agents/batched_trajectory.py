"""BatchedTrajectory -- per-environment timestep assembler (API of rlgym_ppo/batched_agents/batched_trajectory.py).

One instance follows one environment (match); a timestep is complete once the learner has filled in
state/action/log_prob (at send time) and reward/next_state/done/truncated (at receive time).  `get_all()` splits the
stored match timesteps into one trajectory per agent, as seven parallel lists."""
import numpy as np

_FIELDS = ("state", "action", "log_prob", "reward", "next_state", "done", "truncated")


class BatchedTrajectory(object):
    def __init__(self):
        for f in _FIELDS:
            setattr(self, f, None)
        self.complete_timesteps = []

    def update(self):
        """Bank the pending timestep if all seven fields are present; True when that timestep ended the episode."""
        # (written out field by field: this runs once per environment step of every worker)
        if (self.state is None or self.action is None or self.log_prob is None or self.reward is None or self.next_state is None
                or self.done is None or self.truncated is None):
            return False
        if not isinstance(self.reward, (list, tuple, np.ndarray)):
            self.reward = [self.reward]
        self.complete_timesteps.append((self.state, self.action, self.log_prob, self.reward, self.next_state, self.done, self.truncated))
        ended = bool(self.done)
        # `truncated` keeps its last value, like the reference
        self.state = self.action = self.log_prob = self.reward = self.next_state = self.done = None
        return ended

    def get_all(self):
        steps, self.complete_timesteps = self.complete_timesteps, []
        if not steps:
            return []
        n_agents = len(steps[0][3])
        pad = np.zeros_like(steps[0][0][0])
        out = []
        for i in range(n_agents):
            cols = [[] for _ in range(7)]
            for state, action, log_prob, reward, next_state, done, truncated in steps:
                nxt = next_state[i] if i < len(next_state) else pad  # team size changed across a reset
                for c, v in zip(cols, (state[i], action[i], log_prob[i], reward[i], nxt, done, truncated)):
                    c.append(v)
            out.append(cols)
        return out
