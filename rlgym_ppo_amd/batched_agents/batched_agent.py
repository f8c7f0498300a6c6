"""Environment worker process -- the role and the wire behaviour of rlgym_ppo/batched_agents/batched_agent.py:4-222: owns one
gym-style environment, steps it with the actions the learner sends, resets it at episode ends, and ships float32 observations
back through its slab of the shared array (layouts: comm_consts.py).  Same signature as the reference's worker, so either
side of the protocol can be swapped for the reference's."""
import pickle
import socket
import time

import numpy as np

from . import comm_consts as C


def _as_f32(x):
    a = np.asarray(x, dtype=np.float32)
    return a.reshape(1, -1) if a.ndim == 1 else a


def describe_action_space(space):
    """(n_actions, type code): 0 discrete, 1 multi-discrete, 2 continuous -- the codes PPOLearner switches on
    (reference: batched_agent.py:186-199, ppo_learner.py:34-50).  Duck-typed so gym need not be importable."""
    kind = type(space).__name__
    code = {"MultiDiscrete": 1.0, "Box": 2.0}.get(kind, 0.0)
    n = float(space.n) if hasattr(space, "n") else float(np.prod(space.shape))
    return n, code


class StepSlab:
    """Writer side of one worker's slab: float32[shm_size] starting `shm_offset` BYTES into the shared buffer."""

    def __init__(self, shm_buffer, shm_offset, shm_size):
        self.view = np.frombuffer(shm_buffer, dtype=np.float32, offset=shm_offset, count=shm_size)
        self.size = shm_size

    def write_step(self, prev_n_agents, done, truncated, rewards, metrics, metrics_shape, obs):
        """obs: float32 array as the environment returned it (rank 1 for a one-agent env, rank 2 otherwise); metrics_shape: []
        when the worker has no metrics function, else the shape of what it returned (batched_agent.py:146-151)."""
        metrics = np.asarray(metrics, dtype=np.float32)
        metrics_shape = [float(d) for d in metrics_shape]
        state_shape = [float(d) for d in obs.shape]
        head = [float(prev_n_agents), float(done), float(truncated), float(len(state_shape)), float(len(metrics_shape))]
        parts = [head, metrics_shape, state_shape, [float(r) for r in rewards], metrics.ravel(), obs.ravel()]
        count = sum(len(p) for p in parts)
        assert count <= self.size, "ATTEMPTED TO CREATE AGENT MESSAGE BUFFER LARGER THAN MAXIMUM ALLOWED SIZE"
        o = 0
        for p in parts:
            n = len(p)
            self.view[o:o + n] = p
            o += n
        return count


def batched_agent_process(proc_id, endpoint, shm_buffer, shm_offset, shm_size, seed, render, render_delay):
    """Worker main loop (reference signature, batched_agent.py:4).  `endpoint`: the learner's UDP address; `shm_buffer`: the
    shared float32 array; `shm_offset` (bytes) / `shm_size` (floats): this worker's slab."""
    env = None
    pipe = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    try:
        pipe.bind(("127.0.0.1", 0))
        pipe.sendto(b"0", endpoint)  # hello: the learner learns our endpoint from the datagram's source address
        metrics_fn = None
        while env is None:
            data = pickle.loads(pipe.recv(65536))
            if data[0] == C.INIT_TAG:
                build_env_fn, metrics_fn = data[1], data[2]
                env = build_env_fn()
        if hasattr(env.action_space, "seed"):
            env.action_space.seed(seed)
        obs = np.asarray(env.reset(), dtype=np.float32)
        shape = [float(d) for d in obs.shape]
        n_agents = int(shape[0]) if len(shape) > 1 else 1
        pipe.sendto(C.pack_message(C.ENV_RESET_STATE_HEADER + [float(len(shape))] + shape) + obs.tobytes(), endpoint)

        slab = StepSlab(shm_buffer, shm_offset, shm_size)
        step_header = C.pack_message(C.ENV_STEP_DATA_HEADER)
        action_buffer = None
        last_render = time.time()
        while True:
            message = np.frombuffer(pipe.recv(C.WORKER_RECV_SIZE), dtype=np.float32)
            if message.size < C.HEADER_LEN:
                continue
            kind = float(message[0])
            if kind == C.POLICY_ACTIONS_HEADER[0]:
                prev_n_agents = n_agents
                # What the environment is handed is part of the observable behaviour (its arithmetic follows the dtype): like the
                # reference's worker, a float32 copy of the first message refilled in place, replaced by a float64 zeros array
                # after every episode end (batched_agent.py:112-120,140)
                data = message[C.HEADER_LEN:]
                if action_buffer is None:
                    action_buffer = data.reshape(n_agents, -1).copy()
                else:
                    action_buffer[...] = data.reshape(action_buffer.shape)
                step = env.step(action_buffer)
                if len(step) == 4:
                    obs, rew, done, info = step
                    truncated = False
                else:
                    obs, rew, done, truncated, info = step
                rew = [float(rew)] if np.ndim(rew) == 0 else [float(r) for r in rew]
                if done or truncated:
                    obs = env.reset()
                obs = np.asarray(obs, dtype=np.float32)
                if done or truncated:
                    n_agents = int(obs.shape[0]) if obs.ndim > 1 else 1
                    action_buffer = np.zeros((n_agents, action_buffer.shape[-1]))
                if metrics_fn is not None:
                    metrics = np.asarray(metrics_fn(info["state"]))
                    metrics_shape = metrics.shape
                else:
                    metrics, metrics_shape = np.empty((0,), np.float32), ()
                slab.write_step(prev_n_agents, 1.0 if done else 0.0, 1.0 if truncated else 0.0, rew, metrics, metrics_shape, obs)
                pipe.sendto(step_header, endpoint)
                if render:
                    env.render()
                    if render_delay:
                        wait = render_delay - (time.time() - last_render)
                        if wait > 0:
                            time.sleep(wait)
                        last_render = time.time()
            elif kind == C.ENV_SHAPES_HEADER[0]:
                n_acts, code = describe_action_space(env.action_space)
                pipe.sendto(C.pack_message(C.ENV_SHAPES_HEADER + [float(np.prod(env.observation_space.shape)), n_acts, code]), endpoint)
            elif kind == C.STOP_MESSAGE_HEADER[0]:
                break
    except (EOFError, KeyboardInterrupt):
        pass
    except Exception:
        import traceback
        print("ERROR IN BATCHED AGENT LOOP")
        traceback.print_exc()
    finally:
        try:
            pipe.close()
            if env is not None and hasattr(env, "close"):
                env.close()
        except Exception:
            pass
