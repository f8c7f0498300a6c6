"""Wire format of the learner <-> env-worker protocol, as the reference defines it (rlgym_ppo/batched_agents/comm_consts.py:3-15):
every datagram starts with a header of three magic float32 values that names the message kind; the rest of the datagram --
and, for step data, a per-worker slab of a shared float32 array -- carries float32 values in the layouts below.  A worker
process built for the reference and this manager (or the other way round) interoperate because both sides speak exactly this.

  learner -> worker (UDP, 127.0.0.1)
    pickle(("initialization_data", build_env_fn, metrics_fn))      the only non-float message (batched_agent.py:74-80)
    ENV_SHAPES_HEADER                                              request for the environment's shapes
    POLICY_ACTIONS_HEADER + actions[n_agents * k]                  row-major float32 (batched_agent_manager.py:214)
    STOP_MESSAGE_HEADER
  worker -> learner
    b"0"                                                           hello: tells the learner the worker's UDP endpoint
    ENV_RESET_STATE_HEADER + [rank, *shape] + obs.tobytes()        initial observation (batched_agent.py:91-102)
    ENV_SHAPES_HEADER + [obs_size, n_actions, action_space_type]   type 0 discrete / 1 multi-discrete / 2 continuous
    ENV_STEP_DATA_HEADER                                           "the slab holds a new step" (12 bytes; batched_agent.py:166)
  shared slab of worker i = float32[shm_size] at offset i * shm_size (RawArray('f'), batched_agent_manager.py:436-440):
    [prev_n_agents, done, truncated, rank(state), rank(metrics), *metrics_shape, *state_shape, *rewards[prev_n_agents],
     *metrics.ravel(), *obs.ravel()]                               (batched_agent.py:154-164)
"""
import struct

import numpy as np

HEADER_LEN = 3
ENV_SHAPES_HEADER = [82772., 83273., 83774.]
ENV_RESET_STATE_HEADER = [83744., 83774., 83876.]
ENV_STEP_DATA_HEADER = [83775., 53776., 83727.]
POLICY_ACTIONS_HEADER = [12782., 83783., 80784.]
PROC_MESSAGE_SHAPES_HEADER = [63776., 83777., 83778.]
STOP_MESSAGE_HEADER = [11781., 83782., 83983.]

INIT_TAG = "initialization_data"
PACKET_MAX_SIZE = 8192   # largest datagram the reference reads (batched_agent_manager.py:33); workers read 4096
WORKER_RECV_SIZE = 4096


def pack_message(message_floats):
    return struct.pack("%sf" % len(message_floats), *message_floats)


def unpack_message(message_bytes):
    return list(struct.unpack("%sf" % (len(message_bytes) // 4), message_bytes))


def header_of(message_bytes):
    """The three header floats of a datagram (None if it is too short to have one)."""
    if len(message_bytes) < 4 * HEADER_LEN:
        return None
    return np.frombuffer(message_bytes, dtype=np.float32, count=HEADER_LEN).tolist()


def step_slab_floats(prev_n_agents, n_agents, obs_dim, n_metrics=0, metrics_rank=0):
    """Number of float32 values one step occupies in a worker's slab: what `shm_buffer_size // 4` has to cover."""
    return 5 + metrics_rank + 2 + prev_n_agents + n_metrics + n_agents * obs_dim
