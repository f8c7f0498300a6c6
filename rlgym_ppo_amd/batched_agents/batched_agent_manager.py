"""BatchedAgentManager -- rollout front-end with the API of rlgym_ppo/batched_agents/batched_agent_manager.py.

On the accelerated path: `_send_actions` stacks the ready observations into one [n, d] matrix and makes ONE call to
`policy.get_action` (librlppo's fused forward + sampling), and observations are standardised with the reference's
scalar statistics (quirk Q5).  Everything else here is CPU control plane kept interface- AND wire-compatible: env workers are
separate processes that speak the reference's protocol -- UDP datagrams with three-float magic headers plus one slab per
worker of a shared RawArray('f') for the step data (comm_consts.py; reference batched_agent_manager.py:254-299,436-476) -- so a
worker process built for the reference can serve this manager and vice versa.  With n_processes=0 one environment runs inside
the learner process (no IPC), and VectorAgentManager keeps a vectorised environment's rollout on the GPU.
"""
import ctypes
import multiprocessing as mp
import multiprocessing.sharedctypes
import pickle
import selectors
import socket
import time

import numpy as np
import torch

from .. import _native as N
from ..util import WelfordRunningStat
from . import comm_consts as C
from .batched_agent import _as_f32, batched_agent_process, describe_action_space
from .batched_trajectory import BatchedTrajectory

# messages as the manager's logic sees them, whatever carried them
RESET_STATE, STEP_DATA, ENV_SHAPES = "env_reset_state", "env_step_data", "env_shapes"


def parse_step_slab(shm_view):
    """One step out of a worker's slab (layout: comm_consts.py; reference batched_agent_manager.py:254-299) ->
    (prev_n_agents, done, truncated, rewards list, metrics array, observation [n_agents, d] float32 copy)."""
    # (a handful of bulk .tolist() reads instead of ~20 scalar indexings and two np.prod calls: this runs once per worker and step)
    h = shm_view[:5].tolist()
    prev_n, done, truncated, state_rank, metrics_rank = int(h[0]), h[1], h[2], int(h[3]), int(h[4])
    o = 5
    dims = shm_view[o:o + metrics_rank + state_rank].tolist()
    metrics_shape = [int(d) for d in dims[:metrics_rank]]
    state_shape = [int(d) for d in dims[metrics_rank:]]
    o += metrics_rank + state_rank
    if state_rank == 1:
        state_shape = [1, state_shape[0]]
    rews = shm_view[o:o + prev_n].tolist()
    o += prev_n
    n_metrics = 0
    if metrics_rank:
        n_metrics = 1
        for d in metrics_shape:
            n_metrics *= d
    metrics = shm_view[o:o + n_metrics].copy().reshape(metrics_shape if metrics_rank else (0,))
    o += n_metrics
    n_obs = 1
    for d in state_shape:
        n_obs *= d
    obs = shm_view[o:o + n_obs].copy().reshape(state_shape)
    return prev_n, done, truncated, rews, metrics, obs


class _ProcessWorker:
    """Learner-side end of one worker process: a UDP socket bound to 127.0.0.1 and the worker's slab of the shared array."""

    def __init__(self, proc_id, ctx, shm_buffer, shm_size, seed, render, render_delay, target):
        self.sock = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        self.sock.bind(("127.0.0.1", 0))
        offset = proc_id * shm_size * 4
        self.proc = ctx.Process(target=target, args=(proc_id, self.sock.getsockname(), shm_buffer, offset, shm_size, seed, render,
                                                      render_delay), daemon=True)
        self.proc.start()
        self.shm_view = np.frombuffer(shm_buffer, dtype=np.float32, offset=offset, count=shm_size)
        self.child = None
        self._actions_header = C.pack_message(C.POLICY_ACTIONS_HEADER)

    def handshake(self, build_env_fn, metrics_fn):
        _, self.child = self.sock.recvfrom(1)  # the worker's hello datagram: its source address is its endpoint
        self.sock.sendto(pickle.dumps((C.INIT_TAG, build_env_fn, metrics_fn)), self.child)

    def send_actions(self, actions):
        self.sock.sendto(self._actions_header + np.ascontiguousarray(actions, dtype=np.float32).tobytes(), self.child)

    def request_shapes(self):
        self.sock.sendto(C.pack_message(C.ENV_SHAPES_HEADER), self.child)

    def recv(self):
        """Next message from the worker as a tuple, or None for a datagram that is not one of the protocol's."""
        data = self.sock.recv(C.PACKET_MAX_SIZE)
        header = C.header_of(data)
        if header is None:
            return None
        if header[0] == C.ENV_STEP_DATA_HEADER[0]:
            return (STEP_DATA,) + parse_step_slab(self.shm_view)
        floats = np.frombuffer(data, dtype=np.float32)
        if header == C.ENV_RESET_STATE_HEADER:
            rank = int(floats[C.HEADER_LEN])
            shape = [int(d) for d in floats[C.HEADER_LEN + 1:C.HEADER_LEN + 1 + rank]]
            if rank == 1:
                shape = [1, shape[0]]
            return (RESET_STATE, np.array(floats[C.HEADER_LEN + 1 + rank:], dtype=np.float32).reshape(shape))
        if header == C.ENV_SHAPES_HEADER:
            return (ENV_SHAPES,) + tuple(float(x) for x in floats[C.HEADER_LEN:C.HEADER_LEN + 3])
        return None

    def fileno(self):
        return self.sock.fileno()

    def stop(self):
        try:
            if self.child is not None:
                self.sock.sendto(C.pack_message(C.STOP_MESSAGE_HEADER), self.child)
            self.proc.join(timeout=5)
        finally:
            self.sock.close()


class _LocalWorker:
    """n_processes=0: the environment lives in the learner process (useful for vectorised / synthetic envs); same messages,
    no transport."""

    def __init__(self, build_env_fn, metrics_fn, seed):
        self.env = build_env_fn()
        self.metrics_fn = metrics_fn
        if hasattr(self.env.action_space, "seed"):
            self.env.action_space.seed(seed)
        self.obs = _as_f32(self.env.reset())
        self.inbox = [(RESET_STATE, self.obs)]

    def send_actions(self, actions):
        prev_n = self.obs.shape[0]
        step = self.env.step(np.asarray(actions).reshape(prev_n, -1))
        nxt, rew, done, truncated, info = step if len(step) == 5 else (step[0], step[1], step[2], False, step[3])
        rew = [float(rew)] if np.ndim(rew) == 0 else [float(r) for r in rew]
        if done or truncated:
            nxt = self.env.reset()
        self.obs = _as_f32(nxt)
        metrics = self.metrics_fn(info["state"]) if self.metrics_fn is not None else np.empty((0,), np.float32)
        self.inbox.append((STEP_DATA, prev_n, 1.0 if done else 0.0, 1.0 if truncated else 0.0, rew, metrics, self.obs))

    def request_shapes(self):
        n_acts, code = describe_action_space(self.env.action_space)
        self.inbox.append((ENV_SHAPES, float(np.prod(self.env.observation_space.shape)), n_acts, code))

    def recv(self):
        return self.inbox.pop(0)

    def poll(self):
        return bool(self.inbox)

    def stop(self):
        if hasattr(self.env, "close"):
            self.env.close()


class BatchedAgentManager(object):
    def __init__(self, policy, min_inference_size=8, seed=123, standardize_obs=True, steps_per_obs_stats_increment=5):
        self.policy = policy
        self.seed = seed
        self.processes = []
        self.next_obs, self.current_obs, self.current_pids = [], [], []
        self._average_reward = None
        # [r6] the per-message half of the collection loop in C++ (csrc/collector.cpp, rlppo_collector_*): used for worker PROCESSES
        # whose observation statistics are float32 (always, unless they were restored from JSON); False keeps the Python loop below,
        # which is the readable statement of the same behaviour and what an in-process environment (n_processes = 0) runs
        self.native_collect = True
        self._native = None
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

    @property
    def average_reward(self):
        if self._native is not None:
            v, none = ctypes.c_double(0.0), ctypes.c_int32(1)
            N.check(N.lib().rlppo_collector_average_reward(self._native, 0, ctypes.byref(v), ctypes.byref(none)))
            return None if none.value else v.value
        return self._average_reward

    @average_reward.setter
    def average_reward(self, value):
        self._average_reward = value
        if self._native is not None:
            v, none = ctypes.c_double(0.0 if value is None else float(value)), ctypes.c_int32(1 if value is None else 0)
            N.check(N.lib().rlppo_collector_average_reward(self._native, 1, ctypes.byref(v), ctypes.byref(none)))

    # --------------------------------------------------------------------------------------------- set-up
    def init_processes(self, n_processes, build_env_fn, collect_metrics_fn=None, spawn_delay=None, render=False,
                       render_delay=None, shm_buffer_size=8192, worker_target=None):
        """Spawns the env workers (reference signature, batched_agent_manager.py:398-406) and returns (obs size, n actions,
        action space type).  `shm_buffer_size`: BYTES of shared memory per worker for one step's data.  `worker_target`
        (not in the reference): the process entry point -- any function with the reference worker's signature and wire
        behaviour, e.g. the reference's own batched_agent_process."""
        self.n_procs = max(1, n_processes)
        n = self.n_procs
        self.ep_rews = [[0] for _ in range(n)]
        self.trajectory_map = [BatchedTrajectory() for _ in range(n)]
        self.current_obs = [None] * n
        self.next_obs = [None] * n
        self.selector = selectors.DefaultSelector()
        if n_processes <= 0:
            self.processes = [_LocalWorker(build_env_fn, collect_metrics_fn, self.seed)]
        else:
            methods = mp.get_all_start_methods()
            ctx = mp.get_context("forkserver" if "forkserver" in methods else "spawn")
            self.shm_size = shm_buffer_size // 4
            self.shm_buffer = multiprocessing.sharedctypes.RawArray("f", n * self.shm_size)
            target = worker_target or batched_agent_process
            self.processes = []
            for pid in range(n):
                w = _ProcessWorker(pid, ctx, self.shm_buffer, self.shm_size, self.seed + pid, pid == 0 and render, render_delay, target)
                self.selector.register(w.sock, selectors.EVENT_READ, pid)
                self.processes.append(w)
            for w in self.processes:
                w.handshake(build_env_fn, collect_metrics_fn)
                if spawn_delay is not None:
                    time.sleep(spawn_delay)
        self._get_initial_states()
        return self._get_env_shapes()

    def _get_initial_states(self):
        self.current_pids = []
        for pid, w in enumerate(self.processes):
            msg = w.recv()
            while msg is None or msg[0] != RESET_STATE:
                msg = w.recv()
            obs = msg[1]
            if self.standardize_obs:
                if self.obs_stats is None:
                    self.obs_stats = WelfordRunningStat(shape=obs.shape[-1])
                self.obs_stats.increment(obs, obs.shape[0])
            self.current_obs[pid] = obs
            self.current_pids.append(pid)

    def _get_env_shapes(self):
        w = self.processes[0]
        w.request_shapes()
        while True:
            msg = w.recv()
            if msg is not None and msg[0] == ENV_SHAPES:
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
        # (numpy rows from here on: a torch slice per environment and a 0-d tensor per agent and step cost the collector a fifth of
        # its wall clock -- np.asarray over 50,000 0-d tensors alone 0.26 s of 1.1 s, tools/profile_process_collect.py)
        log_probs = log_probs.numpy() if isinstance(log_probs, torch.Tensor) else np.asarray(log_probs)
        step = 0
        for pid, o in zip(ready, obs):
            stop = step + o.shape[0]
            traj = self.trajectory_map[pid]
            traj.state, traj.action, traj.log_prob = inference_batch[step:stop], actions[step:stop], log_probs[step:stop]
            self.processes[pid].send_actions(actions[step:stop])
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
        local = isinstance(self.processes[0], _LocalWorker)
        while n_collected < n_obs_per_inference:
            if local:
                if not self.processes[0].poll():
                    break
                ready = [0]
            else:
                ready = [key.data for key, event in self.selector.select() if event & selectors.EVENT_READ]
            for pid in ready:
                n_collected += self._collect_response(pid, self.processes[pid], collected_metrics, mean0, std0)
        return collected_metrics, n_collected

    def _collect_response(self, pid, worker, collected_metrics, mean0, std0):
        msg = worker.recv()
        if msg is None or msg[0] != STEP_DATA:
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
        if self._native_ok():
            return self._collect_timesteps_native(n)
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

    # ----------------------------------------------------------------------------------- native collection
    def _stats_native_ok(self):
        st = self.obs_stats
        if not self.standardize_obs:
            return True
        if st is None:
            return False
        arrs = (st.running_mean, st.running_variance)
        # float32 (a fresh run) or float64 (restored from JSON: running_stats.py:120-125 -- numpy then standardises in float64)
        return all(isinstance(a, np.ndarray) and a.flags.c_contiguous and a.ndim == 1 for a in arrs) and \
            arrs[0].dtype == arrs[1].dtype and arrs[0].dtype in (np.float32, np.float64) and isinstance(st.count, (int, np.integer))

    def _native_ok(self):
        if self._native is not None:
            if not self._stats_native_ok():
                raise RuntimeError("BatchedAgentManager: the observation statistics are no longer 1-D float32 / float64 arrays of one dtype "
                                   "(the native collector advances them in place); set agent.native_collect = False before the first collect")
            return True
        if not (self.native_collect and self.processes and all(isinstance(w, _ProcessWorker) for w in self.processes)):
            return False
        if not self._stats_native_ok() or any(o is not None and np.ndim(o) != 2 for o in self.current_obs):
            return False
        obs = next((o for o in self.current_obs if o is not None), None)
        if obs is None:
            return False
        # hand the state of the Python loop (the handshake's reset states, who waits for actions) over to the C++ one
        n = len(self.processes)
        fds = (ctypes.c_int32 * n)(*[w.sock.fileno() for w in self.processes])
        ports = (ctypes.c_int32 * n)(*[int(w.child[1]) for w in self.processes])
        h = ctypes.c_void_p()
        self._nat_d = int(obs.shape[1])
        N.check(N.lib().rlppo_collector_create(n, fds, ports, ctypes.c_void_p(ctypes.addressof(self.shm_buffer)), self.shm_size, self._nat_d,
                                               ctypes.byref(h)))
        self._native = h
        for pid, o in enumerate(self.current_obs):
            if o is not None and pid not in self.current_pids:
                a = np.ascontiguousarray(o, dtype=np.float32)
                N.check(N.lib().rlppo_collector_set_obs(h, pid, a.ctypes.data, a.shape[0], 0))
        for pid in self.current_pids:   # (in the order the Python loop would serve them)
            a = np.ascontiguousarray(self.current_obs[pid], dtype=np.float32)
            N.check(N.lib().rlppo_collector_set_obs(h, pid, a.ctypes.data, a.shape[0], 1))
        self.average_reward = self._average_reward
        self._nat_obs = np.zeros((max(64, 8 * n), self._nat_d), dtype=np.float32)
        self._nat_act_shape = None
        return True

    @torch.no_grad()
    def _collect_timesteps_native(self, n):
        """collect_timesteps with the per-message work in C++ (csrc/collector.cpp): per inference ready -> policy.get_action -> send ->
        collect; same results as the Python loop above, value for value (tests/test_native_collector.py)."""
        t1 = time.perf_counter()
        L, h = N.lib(), self._native
        n_collected = 0
        n_obs_per_inference = min(self.min_inference_size, max(1, len(self.processes)))
        rows, got = ctypes.c_int64(0), ctypes.c_int64(0)
        st = self.obs_stats
        count, since = ctypes.c_int64(0), ctypes.c_int64(0)
        one = np.ones(1, dtype=np.float32)
        stats_key, cached = None, None
        while n_collected < n:
            rc = L.rlppo_collector_ready(h, self._nat_obs.ctypes.data, self._nat_obs.shape[0], ctypes.byref(rows))
            if rc == 1002:  # more waiting agents than the staging matrix holds: grow it
                self._nat_obs = np.zeros((2 * self._nat_obs.shape[0], self._nat_d), dtype=np.float32)
                continue
            N.check(rc)
            if rows.value:
                actions, log_probs = self.policy.get_action(self._nat_obs[:rows.value])
                a = np.ascontiguousarray(actions.numpy() if isinstance(actions, torch.Tensor) else actions, dtype=np.float32)
                lp = np.ascontiguousarray(log_probs.numpy() if isinstance(log_probs, torch.Tensor) else log_probs, dtype=np.float32)
                self._nat_act_shape = a.shape[1:]
                N.check(L.rlppo_collector_send(h, a.ctypes.data, max(1, a.size // rows.value), lp.ctypes.data))
            mode, mean, std, f64 = 0, one, one, 0
            if self.standardize_obs:
                key = (int(st.count), self.per_feature_obs_standardization, id(st.running_mean))
                if key != stats_key:   # (mean / std move only when the statistics advanced: every ~6th message)
                    stats_key = key
                    f64 = int(st.running_mean.dtype == np.float64)
                    dt = np.float64 if f64 else np.float32
                    if self.per_feature_obs_standardization:
                        bc = lambda a: np.ascontiguousarray(np.broadcast_to(np.asarray(a).reshape(-1), (self._nat_d,)), dt)  # (fewer than two samples: a (1,) constant)
                        cached = (2, bc(st.mean), bc(st.std), f64)
                    else:   # the scalars of feature 0 (quirk Q5), as they stand when the wait begins
                        cached = (1, np.asarray([st.mean[0]], dt), np.asarray([st.std[0]], dt), f64)
                mode, mean, std, f64 = cached
                count.value, since.value = int(st.count), int(self.steps_since_obs_stats_update)
            want, resume = n_obs_per_inference, 0
            while True:
                rc = L.rlppo_collector_collect(h, want, resume, mode, mean.ctypes.data, std.ctypes.data,
                                               st.running_mean.ctypes.data if mode else None, st.running_variance.ctypes.data if mode else None,
                                               ctypes.byref(count), f64, int(self.steps_per_obs_stats_increment), ctypes.byref(since), ctypes.byref(got))
                if mode:
                    st.count, self.steps_since_obs_stats_update = int(count.value), int(since.value)
                n_collected += got.value
                if rc != 1004:
                    break
                # a signal interrupted the wait; python's handlers have run by now (a KeyboardInterrupt never gets here): wait on
                want, resume = max(1, want - got.value), 1
            if rc == 1003:
                raise TimeoutError("BatchedAgentManager: no worker message for a minute")
            N.check(rc)
        n_steps, aw, n_met, met_floats = ctypes.c_int64(0), ctypes.c_int32(0), ctypes.c_int64(0), ctypes.c_int64(0)
        N.check(L.rlppo_collector_finish(h, ctypes.byref(n_steps), ctypes.byref(aw), ctypes.byref(n_met), ctypes.byref(met_floats)))
        k, d, w = int(n_steps.value), self._nat_d, max(1, int(aw.value))
        states, nxt = np.empty((k, d), np.float32), np.empty((k, d), np.float32)
        actions, logp = np.empty((k, w), np.float32), np.empty(k, np.float32)
        rewards, dones, trunc = np.empty(k, np.float64), np.empty(k, np.float64), np.empty(k, np.float64)
        mvals, mshapes = np.empty(int(met_floats.value), np.float32), np.zeros((int(n_met.value), 9), np.int32)
        N.check(L.rlppo_collector_emit(h, states.ctypes.data, actions.ctypes.data, logp.ctypes.data, rewards.ctypes.data, nxt.ctypes.data,
                                       dones.ctypes.data, trunc.ctypes.data, mvals.ctypes.data, mshapes.ctypes.data))
        metrics, o = [], 0
        for rec in mshapes:
            shape = tuple(int(x) for x in rec[1:1 + rec[0]]) if rec[0] else (0,)
            size = int(np.prod(shape)) if rec[0] else 0
            metrics.append(mvals[o:o + size].copy().reshape(shape))
            o += size
        if k == 0:
            cols = tuple(np.asarray([]) for _ in range(7))
        else:
            cols = (states, actions.reshape((k,) + tuple(self._nat_act_shape or ())), logp, rewards, nxt, dones, trunc)
        self.cumulative_timesteps += n_collected
        return cols, metrics, n_collected, time.perf_counter() - t1

    def cleanup(self):
        if self._native is not None:
            self._average_reward = self.average_reward
            N.lib().rlppo_collector_destroy(self._native)
            self._native = None
        for w in self.processes:
            try:
                w.stop()
            except Exception:
                import traceback
                print("Unable to join process")
                traceback.print_exc()
        self.processes = []
        if getattr(self, "selector", None) is not None:
            self.selector.close()
            self.selector = None
