"""Message tags of the learner <-> env-worker protocol.  The reference tags its UDP datagrams with triples of magic
floats (rlgym_ppo/batched_agents/comm_consts.py); this build moves the same message kinds over multiprocessing pipes
(the IPC is outside the accelerated path), so plain strings suffice."""
INIT = "initialization_data"
RESET_STATE = "env_reset_state"
STEP_DATA = "env_step_data"
POLICY_ACTIONS = "policy_actions"
ENV_SHAPES = "env_shapes"
STOP = "stop"
