"""oracle/host.py -- TEST INFRASTRUCTURE (see oracle/__init__.py).

numpy restatement of the two host-side data structures that feed the hot path:

    fifo_append / shuffled_batches   ExperienceBuffer._cat and get_all_batches_shuffled
                                     (reference: rlgym_ppo/ppo/experience_buffer.py:18-37, 89-102)
    Welford                          WelfordRunningStat (reference: rlgym_ppo/util/running_stats.py:15-137)
    lockstep_rollout                 what BatchedAgentManager.collect_timesteps + BatchedTrajectory assemble
                                     (reference: batched_agent_manager.py:126-172, 230-235, 303-315, 366-384;
                                     batched_trajectory.py:58-105) when every agent is its own one-agent environment
                                     process and all processes answer every step -- the order the product's
                                     VectorAgentManager keeps on the device

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


def lockstep_rollout(reset_obs, step_fn, act_fn, n_steps, standardize=True, stats_every=5, state=None, per_feature=False):
    """-> ((states, actions, log_probs, rewards, next_states, dones, truncated), state).

    reset_obs [n, d]: observations returned by the reset (first call only); step_fn(actions [n, k]) ->
    (obs, rewards, dones, truncated) with auto-reset; act_fn(obs [n, d]) -> (actions [n, k], log_probs [n]).
    `state` carries (next observation as the policy will see it, statistics, cadence counter) from call to call."""
    if state is None:
        stats = None
        if standardize:
            stats = Welford(reset_obs.shape[-1])
            stats.increment(reset_obs, reset_obs.shape[0])   # reset observations enter the statistics ...
        state = dict(cur=np.asarray(reset_obs, np.float32), stats=stats, since=0)   # ... and are acted on raw
    cur, stats = state["cur"], state["stats"]
    n = cur.shape[0]
    S, A, LP, R, NX, D, TR = ([[] for _ in range(n)] for _ in range(7))
    for t in range(n_steps):
        acts, logp = act_fn(cur)
        obs, rew, done, trunc = step_fn(acts)
        obs = np.asarray(obs, np.float32)
        if standardize:
            mean0, std0 = stats.mean[0], stats.std[0]          # fetched before this step's increment
            if per_feature:                                     # the corrected form (not the reference's): own statistics
                mean0, std0 = stats.mean.reshape(-1).copy(), stats.std.reshape(-1).copy()
            if state["since"] > stats_every:
                stats.increment(obs, obs.shape[0])
                state["since"] = 0
            else:
                state["since"] += 1
            nxt = np.clip((obs - mean0) / std0, -5, 5).astype(np.float32)   # scalar statistics of feature 0 (quirk Q5)
        else:
            nxt = obs
        for a in range(n):
            S[a].append(cur[a]); A[a].append(np.asarray(acts[a], np.float32).reshape(-1)); LP[a].append(logp[a])
            R[a].append(rew[a]); NX[a].append(nxt[a]); D[a].append(float(done[a])); TR[a].append(float(trunc[a]))
        cur = nxt
    for a in range(n):                                          # flush: unfinished trajectories are force-truncated
        TR[a][-1] = 1.0 if D[a][-1] == 0 else 0.0
    state["cur"] = cur
    cat = lambda cols: np.asarray([x for a in range(n) for x in cols[a]], np.float32)
    return (cat(S), cat(A), cat(LP), cat(R), cat(NX), cat(D), cat(TR)), state
