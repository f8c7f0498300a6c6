"""Environment worker process (role of rlgym_ppo/batched_agents/batched_agent.py): owns one gym-style environment,
steps it with the actions the learner sends, resets it at episode ends, and ships float32 observations back."""
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


def batched_agent_process(proc_id, conn, seed, render, render_delay):
    env = None
    try:
        tag, build_env_fn, metrics_fn = conn.recv()
        assert tag == C.INIT
        env = build_env_fn()
        if hasattr(env.action_space, "seed"):
            env.action_space.seed(seed)
        obs = _as_f32(env.reset())
        conn.send((C.RESET_STATE, obs))
        while True:
            msg = conn.recv()
            if msg[0] == C.POLICY_ACTIONS:
                prev_n = obs.shape[0]
                step = env.step(np.asarray(msg[1]).reshape(prev_n, -1))
                if len(step) == 4:
                    nxt, rew, done, info = step
                    truncated = False
                else:
                    nxt, rew, done, truncated, info = step
                rew = [float(rew)] if np.ndim(rew) == 0 else [float(r) for r in rew]
                if done or truncated:
                    nxt = env.reset()
                obs = _as_f32(nxt)
                metrics = metrics_fn(info["state"]) if metrics_fn is not None else np.empty((0,), np.float32)
                conn.send((C.STEP_DATA, prev_n, 1.0 if done else 0.0, 1.0 if truncated else 0.0, rew, metrics, obs))
                if render:
                    env.render()
                    if render_delay:
                        time.sleep(render_delay)
            elif msg[0] == C.ENV_SHAPES:
                n_acts, code = describe_action_space(env.action_space)
                conn.send((C.ENV_SHAPES, float(np.prod(env.observation_space.shape)), n_acts, code))
            elif msg[0] == C.STOP:
                break
    except (EOFError, KeyboardInterrupt):
        pass
    except Exception:
        import traceback
        print("ERROR IN BATCHED AGENT LOOP")
        traceback.print_exc()
    finally:
        try:
            conn.close()
            if env is not None and hasattr(env, "close"):
                env.close()
        except Exception:
            pass
