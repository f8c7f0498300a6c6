"""Rollout collection for ONE vectorised environment that steps all of its agents in lockstep, with the rollout kept on
the GPU (device-resident rollout storage, vector-env fast path).

The reference collects through one OS process per environment, a UDP datagram per step and Python lists per agent
(batched_agent_manager.py:126-299, batched_trajectory.py:58-105); at 4096 agents that is 4096 datagrams per step and,
per iteration, seven `np.asarray` flattens plus a 224 MB upload of the states.  Here the policy's padded input rows of
step t are written straight into their final, trajectory-major position on the device (row a*T + t of agent a), so
`Learner.add_new_experience` runs the value pass, the GAE scan and the buffer submit on them without a copy through the host.

What comes out is exactly what the reference's assembler would produce for the same interaction sequence when every agent
is its own trajectory stream (tests/test_gpu_vector_rollout.py compares with a numpy restatement of these rules):
  * trajectories are concatenated agent by agent, steps in order (batched_trajectory.py:58-105);
  * the last collected step of every agent is force-marked truncated unless it is terminal (quirk Q4, the flush at
    batched_agent_manager.py:154-172), and the episode continues in the next collect;
  * observations are standardised with the SCALAR statistics of feature 0 and clipped to +-5 (quirk Q5, :303-315), the
    running statistics are advanced every `steps_per_obs_stats_increment` steps with the raw observations (:233-235);
  * `next_states[i]` is the observation the environment returned for step i (after its auto-reset at episode ends).

Environment interface: `reset() -> obs [n, d]`; `step(actions [n, k]) -> (obs [n, d], rewards [n], dones [n],
truncated [n], info)` with auto-reset of finished agents; `observation_space.shape`, `action_space` as in the reference.
"""
import time

import numpy as np
import torch

from ..util.running_stats import WelfordRunningStat
from .batched_agent import describe_action_space


class VectorAgentManager(object):
    def __init__(self, policy, min_inference_size=8, seed=123, standardize_obs=True, steps_per_obs_stats_increment=5):
        self.policy = policy
        self.seed = seed
        self.standardize_obs = standardize_obs
        self.steps_per_obs_stats_increment = steps_per_obs_stats_increment
        self.steps_since_obs_stats_update = 0
        self.per_feature_obs_standardization = False  # True: every feature with its own statistics (not the reference's Q5)
        self.obs_stats = None
        self.cumulative_timesteps = 0
        self.average_reward = None
        self.env = None
        self.collect_metrics_fn = None
        self.n_agents = 0
        self.value_input_rows = None  # [N + 1, ld] device rows of the last collect: states ++ the last next_state
        self._next_rows = None        # padded device rows of the observation the agents act on next
        self._pending_obs = None      # fused collect: (raw device observations, standardisation scalars) the agents act on next
        self._ep_rews = None

    # same signature as BatchedAgentManager.init_processes (learner.py:140-150); n_processes is ignored
    def init_processes(self, n_processes, build_env_fn, collect_metrics_fn=None, spawn_delay=None, render=False,
                       render_delay=None, shm_buffer_size=8192):
        self.env = build_env_fn()
        self.collect_metrics_fn = collect_metrics_fn
        if hasattr(self.env.action_space, "seed"):
            self.env.action_space.seed(self.seed)
        obs = np.asarray(self.env.reset(), dtype=np.float32)
        self.n_agents, d = obs.shape
        self._ep_rews = np.zeros(self.n_agents, np.float64)
        self.obs_stats = None
        if self.standardize_obs:  # the reset observations enter the statistics and are acted on RAW
            self.obs_stats = WelfordRunningStat(shape=d)                      # (batched_agent_manager.py:366-384)
            self.obs_stats.increment(obs, obs.shape[0])
        self._initial_obs = obs
        n_acts, code = describe_action_space(self.env.action_space)
        return int(np.prod(self.env.observation_space.shape)), int(n_acts), int(code)

    def _standardize_scalars(self):
        if not self.standardize_obs:
            return None
        key = (id(self.obs_stats), self.obs_stats.count, id(self.obs_stats.running_mean), self.per_feature_obs_standardization)
        cached = getattr(self, "_scalars_cache", None)
        if cached is not None and cached[0] == key:   # the statistics have not moved since (they move every 6th step)
            return cached[1]
        if self.per_feature_obs_standardization:  # device vectors -> rlppo_pad_rows_per_feature
            out = (torch.from_numpy(np.asarray(self.obs_stats.mean, np.float32).reshape(-1).copy()),
                   torch.from_numpy(np.asarray(self.obs_stats.std, np.float32).reshape(-1).copy()))
        else:
            out = (float(self.obs_stats.mean[0]), float(self.obs_stats.std[0]))
        self._scalars_cache = (key, out)
        return out

    @torch.no_grad()
    def collect_timesteps(self, n):
        """-> ((states, actions, log_probs, rewards, next_states, dones, truncated), metrics, n_collected, seconds):
        device tensors in trajectory-major order; `self.value_input_rows` holds states ++ last next_state contiguously."""
        if getattr(self.policy, "fused_step", False):
            return self._collect_fused(n)
        return self._collect_chain(n)

    def _collect_fused(self, n):
        """[r3] The discrete policy's step is ONE launch (DiscreteFF.step -> rlppo_discrete_step): it standardises and pads the raw
        observations of step t straight into the rollout storage, runs the policy, stores the action (as the buffer's float
        encoding) and the log-probability into the storage and the action indices into pinned host memory for the environment.
        The storage is TIME-major while collecting (everything a step writes is contiguous: no scatter copies) and is transposed
        into the reference's trajectory-major order once per collect.  Same numbers as _collect_chain, bit for bit."""
        t1 = time.perf_counter()
        arena = self.policy.arena
        dev, ld, na = arena.device, arena.ld_in, self.n_agents
        T = max(1, -(-int(n) // na))
        N_ = na * T
        S = torch.empty((T + 1, na, ld), dtype=torch.float32, device=dev)     # S[t] = policy-input rows of step t; S[T]: the next collect's
        acts_tm = torch.empty((T, na), dtype=torch.float32, device=dev)
        logp_tm = torch.empty((T, na), dtype=torch.float32, device=dev)
        # [r4] the host side of a step is kept as small as the launch: rewards / flags are TIME-major too (row t is one contiguous
        # write; round 3 wrote three stride-T columns per step) and transposed on the device once per collect; the raw observations
        # go through a small ring of page-locked staging rows (asynchronous upload: a pageable source is copied synchronously);
        # the standardisation scalars are recomputed only when the statistics moved; the episode-reward average runs over the
        # ended agents' values as Python floats (same operations in the same order, no numpy scalar per agent).
        rdt = np.empty((3, T, na), np.float32)                                # rewards, dones, truncated of step t: rdt[:, t]
        metrics = []
        if self._pending_obs is None:  # first collect: the reset observations are acted on RAW (batched_agent_manager.py:366-384)
            self._pending_obs = (torch.from_numpy(np.ascontiguousarray(self._initial_obs)).to(dev), None)
        obs_dev, scalars = self._pending_obs
        stage = self._obs_staging(na, arena.d_in)
        for t in range(T):
            a_host, _ = self.policy.step(obs_dev, standardize=scalars, rows_out=S[t], actions_f32=acts_tm[t], logp_out=logp_tm[t],
                                         to_host="actions")
            step = self.env.step(a_host.numpy().astype(np.float32).reshape(na, -1))
            if len(step) == 4:
                obs, r, d, info = step
                tr = 0.0
            else:
                obs, r, d, tr, info = step
            pin_t, pin_np = stage[t % len(stage)]   # (reused three steps later: every step ends in a stream synchronisation)
            np.copyto(pin_np, obs, casting="unsafe")
            obs_dev = pin_t.to(dev, non_blocking=True)                         # raw observations: one asynchronous upload per step
            row = rdt[:, t]
            row[0], row[1], row[2] = r, d, tr
            if self.collect_metrics_fn is not None:
                metrics.append(self.collect_metrics_fn(info["state"]))
            scalars = self._standardize_scalars()  # fetched BEFORE this step's increment (batched_agent_manager.py:230-235)
            if self.standardize_obs:  # same cadence as one worker response per step
                if self.steps_since_obs_stats_update > self.steps_per_obs_stats_increment:
                    self._increment_obs_stats(obs_dev)
                    self.steps_since_obs_stats_update = 0
                else:
                    self.steps_since_obs_stats_update += 1
            self._track_rewards(row[0], (row[1] + row[2]) > 0)
        arena.stage_obs(obs_dev, scalars, out=S[T])          # the rows the agents act on next = next_states of the last step
        self._pending_obs = (obs_dev, scalars)
        if getattr(self.policy, "noise_mode", None) == "host" and hasattr(self.policy, "prefetch_noise"):
            # the NEXT collect's T draws of the reference's CPU noise stream, produced on the helper threads while the value pass,
            # the GAE scan and PPOLearner.learn run (none of them touches torch's CPU generator; if anything does, the chain is
            # dropped at the next draw: engine.HostExponential)
            self.policy.prefetch_noise(na, T)
        flat = torch.empty((N_ + 1, ld), dtype=torch.float32, device=dev)
        nxt_flat = torch.empty((N_, ld), dtype=torch.float32, device=dev)
        flat[:N_].view(na, T, ld).copy_(S[:T].transpose(0, 1))               # time-major -> trajectory-major, once
        nxt_flat.view(na, T, ld).copy_(S[1:].transpose(0, 1))
        flat[N_].copy_(S[T][na - 1])                         # next_states[-1]: what add_new_experience appends (learner.py:347)
        rdt[2, T - 1] = np.where(rdt[1, T - 1] == 0, 1.0, 0.0)   # flush rule (quirk Q4)
        rdt_dev = torch.from_numpy(rdt).to(dev).transpose(1, 2).contiguous()   # [3, na, T]: trajectory-major, transposed on the device
        self.value_input_rows = flat
        self._next_rows = S[T]
        self.cumulative_timesteps += N_
        experience = (flat[:N_], acts_tm.t().reshape(N_, 1), logp_tm.t().reshape(N_), rdt_dev[0].reshape(N_), nxt_flat,
                      rdt_dev[1].reshape(N_), rdt_dev[2].reshape(N_))
        return experience, metrics, N_, time.perf_counter() - t1

    def _obs_staging(self, na, d):
        """Three page-locked [na, d] float32 staging buffers (tensor, numpy view), allocated once."""
        st = getattr(self, "_stage", None)
        if st is None or st[0][0].shape != (na, d):
            pin = torch.cuda.is_available()
            ts = [torch.empty((na, d), dtype=torch.float32, pin_memory=pin) for _ in range(3)]
            st = self._stage = [(t, t.numpy()) for t in ts]
        return st

    def _collect_chain(self, n):
        t1 = time.perf_counter()
        arena = self.policy.arena
        dev, ld, na = arena.device, arena.ld_in, self.n_agents
        T = max(1, -(-int(n) // na))
        N_ = na * T
        flat = torch.empty((N_ + 1, ld), dtype=torch.float32, device=dev)
        nxt_flat = torch.empty((N_, ld), dtype=torch.float32, device=dev)
        s3, n3 = flat[:N_].view(na, T, ld), nxt_flat.view(na, T, ld)
        acts = logp = None
        rews = np.empty((na, T), np.float32)
        dones = np.empty((na, T), np.float32)
        trunc = np.empty((na, T), np.float32)
        metrics = []
        if self._next_rows is None:
            self._next_rows = arena.stage_obs(self._initial_obs)
        rows = self._next_rows
        for t in range(T):
            a_dev, lp_dev = self.policy.act_padded(rows)
            if acts is None:
                k = 1 if a_dev.dim() == 1 else a_dev.shape[1]
                acts = torch.empty((na, T, k), dtype=torch.float32, device=dev)
                logp = torch.empty((na, T), dtype=torch.float32, device=dev)
            s3[:, t].copy_(rows)
            acts[:, t].copy_(a_dev.view(na, -1))
            logp[:, t].copy_(lp_dev)
            step = self.env.step(a_dev.cpu().numpy().astype(np.float32).reshape(na, -1))
            if len(step) == 4:
                obs, r, d, info = step
                tr = np.zeros(na, np.float32)
            else:
                obs, r, d, tr, info = step
            obs = np.asarray(obs, dtype=np.float32)
            obs_dev = torch.from_numpy(np.ascontiguousarray(obs)).to(dev)   # raw observations: one upload per step
            rews[:, t], dones[:, t], trunc[:, t] = r, d, tr
            if self.collect_metrics_fn is not None:
                metrics.append(self.collect_metrics_fn(info["state"]))
            scalars = self._standardize_scalars()  # fetched BEFORE this step's increment (batched_agent_manager.py:230-235)
            if self.standardize_obs:  # same cadence as one worker response per step
                if self.steps_since_obs_stats_update > self.steps_per_obs_stats_increment:
                    self._increment_obs_stats(obs_dev)
                    self.steps_since_obs_stats_update = 0
                else:
                    self.steps_since_obs_stats_update += 1
            self._track_rewards(rews[:, t], (dones[:, t] + trunc[:, t]) > 0)
            rows = arena.stage_obs(obs_dev, scalars)
            n3[:, t].copy_(rows)
        flat[N_].copy_(rows[na - 1])                      # next_states[-1]: what add_new_experience appends (learner.py:347)
        trunc[:, T - 1] = np.where(dones[:, T - 1] == 0, 1.0, 0.0)   # flush rule (quirk Q4)
        up = lambda x: torch.from_numpy(np.ascontiguousarray(x.reshape(-1))).to(dev)
        self.value_input_rows = flat
        self._next_rows = rows
        self.cumulative_timesteps += N_
        experience = (flat[:N_], acts.view(N_, -1), logp.view(N_), up(rews), nxt_flat, up(dones), up(trunc))
        return experience, metrics, N_, time.perf_counter() - t1

    def _increment_obs_stats(self, obs_dev):
        """WelfordRunningStat.increment(obs, n) on the device (bit-exact with the host class's sample-by-sample update in the
        state's dtype -- float32, or float64 after a checkpoint load -- which costs ~10 ms of Python per 4096 samples); the host
        object stays the owner of the state (checkpoints)."""
        from ..util import device_stats
        device_stats.increment(self.obs_stats, obs_dev)

    def _track_rewards(self, r, ended):
        """Episode-reward average as batched_agent_manager.py:377-399 keeps it, one agent = one stream: the ended agents' episode
        sums enter the 0.9 / 0.1 average in agent order (Python floats, the reference's operations in the reference's order)."""
        self._ep_rews += r
        idx = np.flatnonzero(ended)
        if idx.size == 0:
            return
        avg = self.average_reward
        for x in self._ep_rews[idx].tolist():
            avg = x if avg is None else avg * 0.9 + x * 0.1
        self.average_reward = avg
        self._ep_rews[idx] = 0.0

    def cleanup(self):
        if self.env is not None and hasattr(self.env, "close"):
            self.env.close()
        self.env = None
