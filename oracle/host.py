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

    reference_layout_worker          the env-worker side of the learner <-> worker wire protocol (reference:
                                     batched_agents/batched_agent.py:4-222, comm_consts.py:3-15): UDP datagrams with
                                     three-float magic headers + one slab of a shared float32 array for the step data.
                                     Pinned byte for byte by tests/golden/g11_wire.npz (captured from the imported
                                     reference worker) and used to show that the product's manager serves a worker that
                                     speaks the reference layout (tests/test_wire_format.py)

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


# ------------------------------------------------------------------------------------------------- wire protocol
WIRE_HEADERS = dict(env_shapes=(82772., 83273., 83774.), reset_state=(83744., 83774., 83876.), step_data=(83775., 53776., 83727.),
                    policy_actions=(12782., 83783., 80784.), stop=(11781., 83782., 83983.))  # comm_consts.py:4-9


def reference_layout_worker(proc_id, endpoint, shm_buffer, shm_offset, shm_size, seed, render, render_delay):
    """What a reference env worker puts on the wire, restated (batched_agent.py:4-222):
      hello b"0" -> wait for pickle(("initialization_data", build_env_fn, metrics_fn)) -> build + seed + reset the env ->
      [reset header, rank, *shape] as packed float32 followed by the raw observation bytes -> then per datagram received:
        policy-actions header + float32 actions: step (reset at episode ends), write
            [prev_n_agents, done, truncated, rank(obs), rank(metrics), *metrics_shape, *obs_shape, *rewards, *metrics, *obs]
            as float32 into the slab (float32[shm_size] at byte offset shm_offset) and send the 12-byte step header;
        env-shapes header: reply header + [prod(obs space shape), n_actions, type code 0/1/2];
        stop header: leave."""
    import pickle
    import socket
    import struct
    f32 = lambda vals: struct.pack("%df" % len(vals), *vals)
    sock = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    sock.bind(("127.0.0.1", 0))
    sock.sendto(b"0", endpoint)
    env = metrics_fn = None
    while env is None:
        msg = pickle.loads(sock.recv(4096))
        if msg[0] == "initialization_data":
            env, metrics_fn = msg[1](), msg[2]
    try:
        env.action_space.seed(seed)
        obs = np.asarray(env.reset(), dtype=np.float32)
        n_agents = obs.shape[0] if obs.ndim > 1 else 1
        sock.sendto(f32(list(WIRE_HEADERS["reset_state"]) + [float(obs.ndim)] + [float(d) for d in obs.shape]) + obs.tobytes(), endpoint)
        slab = np.frombuffer(shm_buffer, dtype=np.float32, offset=shm_offset, count=shm_size)
        act_buf = None
        while True:
            floats = np.frombuffer(sock.recv(4096), dtype=np.float32)
            if floats[0] == WIRE_HEADERS["policy_actions"][0]:
                prev_n = n_agents
                # the action buffer the environment sees: a float32 copy of the first message, refilled in place afterwards;
                # after an episode end it is replaced by np.zeros(...) -- float64 -- so from then on the environment is stepped
                # with float64 actions (batched_agent.py:112-120,140)
                if act_buf is None:
                    act_buf = floats[3:].reshape(int(n_agents), -1).copy()
                else:
                    act_buf[...] = floats[3:].reshape(act_buf.shape)
                out = env.step(act_buf)
                obs, rew, done, truncated = out[0], out[1], out[2], (out[3] if len(out) == 5 else False)
                info = out[-1]
                if n_agents == 1 and not isinstance(rew, list):
                    rew = [float(rew)]
                if done or truncated:
                    obs = np.asarray(env.reset(), dtype=np.float32)
                    n_agents = obs.shape[0] if obs.ndim > 1 else 1
                    act_buf = np.zeros((int(n_agents), act_buf.shape[-1]))
                obs = np.asarray(obs, dtype=np.float32)
                if metrics_fn is not None:
                    metrics = metrics_fn(info["state"])
                    mshape = [float(d) for d in metrics.shape]
                else:
                    metrics, mshape = np.empty((0,)), []
                record = ([float(prev_n), 1.0 if done else 0.0, 1.0 if truncated else 0.0, float(obs.ndim), float(len(mshape))]
                          + mshape + [float(d) for d in obs.shape] + [float(r) for r in rew]
                          + [float(m) for m in np.ravel(metrics)] + [float(x) for x in obs.ravel()])
                assert len(record) <= shm_size
                slab[:len(record)] = record
                sock.sendto(f32(list(WIRE_HEADERS["step_data"])), endpoint)
            elif floats[0] == WIRE_HEADERS["env_shapes"][0]:
                kind = type(env.action_space).__name__
                code = 1.0 if kind == "MultiDiscrete" else (2.0 if kind == "Box" else 0.0)
                n_acts = float(env.action_space.n) if hasattr(env.action_space, "n") else float(np.prod(env.action_space.shape))
                sock.sendto(f32(list(WIRE_HEADERS["env_shapes"]) + [float(np.prod(env.observation_space.shape)), n_acts, code]), endpoint)
            elif floats[0] == WIRE_HEADERS["stop"][0]:
                break
    finally:
        sock.close()
        env.close()


def drive_worker(worker_fn, build_env_fn, metrics_fn, actions, shm_floats=2048, seed=7):
    """Scripted learner side for ONE worker (run in a thread of this process): performs the handshake, asks for the shapes,
    sends every row block of `actions` and records everything the worker produced -> dict(reset, shapes, step_headers, slabs):
    raw datagram bytes (uint8) and, after every step header, a copy of the slab's first `used` floats (`used` parsed from the
    layout).  The fixture tests/golden/g11_wire.npz is this recording of the REFERENCE worker."""
    import multiprocessing.sharedctypes
    import pickle
    import socket
    import struct
    import threading
    parent = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    parent.bind(("127.0.0.1", 0))
    parent.settimeout(20.0)
    shm = multiprocessing.sharedctypes.RawArray("f", 2 * shm_floats)
    offset = shm_floats * 4  # the second slab: a worker that ignores its offset is caught
    th = threading.Thread(target=worker_fn, args=(1, parent.getsockname(), shm, offset, shm_floats, seed, False, None), daemon=True)
    th.start()
    f32 = lambda vals: struct.pack("%df" % len(vals), *vals)
    _, child = parent.recvfrom(1)
    parent.sendto(pickle.dumps(("initialization_data", build_env_fn, metrics_fn)), child)
    rec = dict(reset=np.frombuffer(parent.recv(8192), np.uint8).copy())
    parent.sendto(f32(WIRE_HEADERS["env_shapes"]), child)
    rec["shapes"] = np.frombuffer(parent.recv(8192), np.uint8).copy()
    view = np.frombuffer(shm, dtype=np.float32, offset=offset, count=shm_floats)
    other = np.frombuffer(shm, dtype=np.float32, offset=0, count=shm_floats)
    headers, slabs = [], []
    for a in actions:
        parent.sendto(f32(WIRE_HEADERS["policy_actions"]) + np.ascontiguousarray(a, np.float32).tobytes(), child)
        headers.append(np.frombuffer(parent.recv(8192), np.uint8).copy())
        prev_n, srank, mrank = int(view[0]), int(view[3]), int(view[4])
        mshape = [int(d) for d in view[5:5 + mrank]]
        sshape = [int(d) for d in view[5 + mrank:5 + mrank + srank]]
        used = 5 + mrank + srank + prev_n + (int(np.prod(mshape)) if mrank else 0) + int(np.prod(sshape))
        slabs.append(view[:used].copy())
    parent.sendto(f32(WIRE_HEADERS["stop"]), child)
    th.join(timeout=10)
    assert not th.is_alive() and not other.any(), "worker did not stop / wrote outside its slab"
    parent.close()
    rec["step_headers"], rec["slabs"] = headers, slabs
    return rec
