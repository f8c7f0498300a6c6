"""BatchedAgentManager -- rollout front-end with the API of rlgym_ppo/batched_agents/batched_agent_manager.py.

On the accelerated path: `_send_actions` stacks the ready observations into one [n, d] matrix and makes ONE call to
`policy.get_action` (librlppo's fused forward + sampling), and observations are standardised with the reference's
scalar statistics (quirk Q5).  Everything else here is CPU control plane kept interface-compatible: env workers are
separate processes (or, with n_processes=0, one in-process environment) talking over multiprocessing pipes.
"""
import multiprocessing as mp
import time
from multiprocessing.connection import wait

import numpy as np
import torch

from ..util import WelfordRunningStat
from . import comm_consts as C
from .batched_agent import _as_f32, batched_agent_process, describe_action_space
from .batched_trajectory import BatchedTrajectory


class _LocalWorker:
    """n_processes=0: the environment lives in the learner process (useful for vectorised / synthetic envs)."""

    def __init__(self, build_env_fn, metrics_fn, seed):
        self.env = build_env_fn()
        self.metrics_fn = metrics_fn
        if hasattr(self.env.action_space, "seed"):
            self.env.action_space.seed(seed)
        self.obs = _as_f32(self.env.reset())
        self.inbox = [(C.RESET_STATE, self.obs)]

    def send(self, msg):
        if msg[0] == C.POLICY_ACTIONS:
            prev_n = self.obs.shape[0]
            step = self.env.step(np.asarray(msg[1]).reshape(prev_n, -1))
            nxt, rew, done, truncated, info = step if len(step) == 5 else (step[0], step[1], step[2], False, step[3])
            rew = [float(rew)] if np.ndim(rew) == 0 else [float(r) for r in rew]
            if done or truncated:
                nxt = self.env.reset()
            self.obs = _as_f32(nxt)
            metrics = self.metrics_fn(info["state"]) if self.metrics_fn is not None else np.empty((0,), np.float32)
            self.inbox.append((C.STEP_DATA, prev_n, 1.0 if done else 0.0, 1.0 if truncated else 0.0, rew, metrics, self.obs))
        elif msg[0] == C.ENV_SHAPES:
            n_acts, code = describe_action_space(self.env.action_space)
            self.inbox.append((C.ENV_SHAPES, float(np.prod(self.env.observation_space.shape)), n_acts, code))

    def recv(self):
        return self.inbox.pop(0)

    def poll(self):
        return bool(self.inbox)

    def close(self):
        if hasattr(self.env, "close"):
            self.env.close()


class BatchedAgentManager(object):
    def __init__(self, policy, min_inference_size=8, seed=123, standardize_obs=True, steps_per_obs_stats_increment=5):
        self.policy = policy
        self.seed = seed
        self.processes = []
        self.next_obs, self.current_obs, self.current_pids = [], [], []
        self.average_reward = None
        self.cumulative_timesteps = 0
        self.min_inference_size = min_inference_size
        self.standardize_obs = standardize_obs
        self.steps_per_obs_stats_increment = steps_per_obs_stats_increment
        self.steps_since_obs_stats_update = 0
        self.per_feature_obs_standardization = False  # True: every feature with its own statistics (not the reference's Q5)
        self.obs_stats = None
        self.ep_rews = []
        self.trajectory_map = []
        self.completed_trajectories = []
        self.n_procs = 0

    # --------------------------------------------------------------------------------------------- set-up
    def init_processes(self, n_processes, build_env_fn, collect_metrics_fn=None, spawn_delay=None, render=False,
                       render_delay=None, shm_buffer_size=8192):
        self.n_procs = max(1, n_processes)
        n = self.n_procs
        self.ep_rews = [[0] for _ in range(n)]
        self.trajectory_map = [BatchedTrajectory() for _ in range(n)]
        self.current_obs = [None] * n
        self.next_obs = [None] * n
        if n_processes <= 0:
            self.processes = [(None, _LocalWorker(build_env_fn, collect_metrics_fn, self.seed))]
        else:
            methods = mp.get_all_start_methods()
            ctx = mp.get_context("forkserver" if "forkserver" in methods else "spawn")
            self.processes = []
            for pid in range(n):
                parent, child = ctx.Pipe()
                proc = ctx.Process(target=batched_agent_process,
                                   args=(pid, child, self.seed + pid, pid == 0 and render, render_delay), daemon=True)
                proc.start()
                child.close()
                if spawn_delay is not None:
                    time.sleep(spawn_delay)
                parent.send((C.INIT, build_env_fn, collect_metrics_fn))
                self.processes.append((proc, parent))
        self._get_initial_states()
        return self._get_env_shapes()

    def _get_initial_states(self):
        self.current_pids = []
        for pid, (_, conn) in enumerate(self.processes):
            tag, obs = conn.recv()
            assert tag == C.RESET_STATE
            if self.standardize_obs:
                if self.obs_stats is None:
                    self.obs_stats = WelfordRunningStat(shape=obs.shape[-1])
                self.obs_stats.increment(obs, obs.shape[0])
            self.current_obs[pid] = obs
            self.current_pids.append(pid)

    def _get_env_shapes(self):
        _, conn = self.processes[0]
        conn.send((C.ENV_SHAPES,))
        while True:
            msg = conn.recv()
            if msg[0] == C.ENV_SHAPES:
                return int(msg[1]), int(msg[2]), int(msg[3])

    # ------------------------------------------------------------------------------------------- rollout
    @torch.no_grad()
    def _send_actions(self):
        ready = [pid for pid in self.current_pids if self.current_obs[pid] is not None]
        if not ready:
            return
        obs = [self.current_obs[pid] for pid in ready]
        inference_batch = np.concatenate(obs, axis=0)                  # [n_ready_agents, d]
        actions, log_probs = self.policy.get_action(inference_batch)   # one fused launch sequence for the whole batch
        actions = actions.numpy().astype(np.float32)
        step = 0
        for pid, o in zip(ready, obs):
            stop = step + o.shape[0]
            traj = self.trajectory_map[pid]
            traj.state, traj.action, traj.log_prob = inference_batch[step:stop], actions[step:stop], log_probs[step:stop]
            self.processes[pid][1].send((C.POLICY_ACTIONS, actions[step:stop]))
            step = stop
        self.current_pids = []

    def _collect_responses(self, n_obs_per_inference):
        n_collected = 0
        self.current_pids = []
        collected_metrics = []
        mean0 = std0 = None
        if self.standardize_obs:
            mean0, std0 = self.obs_stats.mean[0], self.obs_stats.std[0]   # scalars of feature 0 (quirk Q5)
            if self.per_feature_obs_standardization:
                mean0, std0 = self.obs_stats.mean.reshape(-1), self.obs_stats.std.reshape(-1)  # broadcast over the rows
        conns = {conn: pid for pid, (_, conn) in enumerate(self.processes)}
        local = isinstance(self.processes[0][1], _LocalWorker)
        while n_collected < n_obs_per_inference:
            ready = [c for c in conns if c.poll()] if local else wait(list(conns))
            if local and not ready:
                break
            for conn in ready:
                n_collected += self._collect_response(conns[conn], conn, collected_metrics, mean0, std0)
        return collected_metrics, n_collected

    def _collect_response(self, pid, conn, collected_metrics, mean0, std0):
        msg = conn.recv()
        if msg[0] != C.STEP_DATA:
            return 0
        _, prev_n, done, truncated, rews, metrics, nxt = msg
        collected_metrics.append(metrics)
        if self.standardize_obs:
            if self.steps_since_obs_stats_update > self.steps_per_obs_stats_increment:
                self.obs_stats.increment(nxt, nxt.shape[0])
                self.steps_since_obs_stats_update = 0
            else:
                self.steps_since_obs_stats_update += 1
            nxt = np.clip((nxt - mean0) / std0, a_min=-5, a_max=5)
        ep = self.ep_rews[pid]
        for i, r in enumerate(rews):
            if i >= len(ep):
                ep.append(r)
            else:
                ep[i] += r
        if done or truncated:
            if self.average_reward is None:
                self.average_reward = ep[0]
            else:
                for r in ep:
                    self.average_reward = self.average_reward * 0.9 + r * 0.1
            self.ep_rews[pid] = [0]
        if pid not in self.current_pids:
            self.current_pids.append(pid)
        self.next_obs[pid] = nxt
        traj = self.trajectory_map[pid]
        traj.reward, traj.next_state, traj.done, traj.truncated = list(rews), nxt, done, truncated
        if nxt.shape[0] != prev_n:  # agent count changed across the reset: start a fresh assembler
            traj.update()
            self.completed_trajectories.append(traj)
            self.trajectory_map[pid] = BatchedTrajectory()
        return prev_n

    def _sync_trajectories(self):
        for pid, traj in enumerate(self.trajectory_map):
            if traj.update():
                self.completed_trajectories.append(traj)
                self.trajectory_map[pid] = BatchedTrajectory()

    def collect_timesteps(self, n):
        """-> ((states, actions, log_probs, rewards, next_states, dones, truncated), metrics, n_collected, seconds),
        trajectory-concatenated, last step of every flushed trajectory force-marked truncated if not done (quirk Q4)."""
        t1 = time.perf_counter()
        cols = [[] for _ in range(7)]
        n_collected = 0
        n_obs_per_inference = min(self.min_inference_size, max(1, len(self.processes)))
        metrics = []
        while n_collected < n:
            self._send_actions()
            m, k = self._collect_responses(n_obs_per_inference)
            n_collected += k
            metrics += m
            for pid in self.current_pids:
                if self.next_obs[pid] is not None:
                    self.current_obs[pid] = self.next_obs[pid]
                    self.next_obs[pid] = None
            self._sync_trajectories()
        for pid, traj in enumerate(self.trajectory_map):
            self.completed_trajectories.append(traj)
            fresh = BatchedTrajectory()
            if traj.state is not None and traj.reward is None:
                # an action is in flight for this worker: keep its (state, action, log_prob) so that the response,
                # which arrives during the next collect_timesteps call, is paired with the step that produced it
                fresh.state, fresh.action, fresh.log_prob = traj.state, traj.action, traj.log_prob
            self.trajectory_map[pid] = fresh
        for traj in self.completed_trajectories:
            for seq in traj.get_all():
                seq[6][-1] = 1 if seq[5][-1] == 0 else 0
                for c, s in zip(cols, seq):
                    c += s
        self.completed_trajectories = []
        self.cumulative_timesteps += n_collected
        return tuple(np.asarray(c) for c in cols), metrics, n_collected, time.perf_counter() - t1

    def cleanup(self):
        for proc, conn in self.processes:
            try:
                if proc is not None:
                    conn.send((C.STOP,))
                    proc.join(timeout=5)
                conn.close()
            except Exception:
                import traceback
                print("Unable to join process")
                traceback.print_exc()
        self.processes = []
