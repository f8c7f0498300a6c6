"""One whole iteration at the cfg2 scale (4096 agents x 128 steps = 524,288 samples): rollout collection with the
device-resident VectorAgentManager, add_new_experience (value pass + GAE + buffer submit) fed from the device rollout vs
fed from host arrays (the reference's hand-over format), and the PPO update.  GPU only."""
import os, sys, time, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import synthetic_env
from rlgym_ppo_amd import Learner

N_AGENTS, T = 4096, 128


class FastVectorEnv(synthetic_env.SyntheticVectorEnv):
    """Pre-drawn observations: the bench is about the learner side, not numpy's randn."""
    def __init__(self):
        super().__init__(n_agents=N_AGENTS, seed=1)
        self._pool = (np.random.RandomState(0).randn(8, N_AGENTS, 107) * 2 + 0.5).astype(np.float32)
        self._i = 0
    def _obs(self):
        self._i += 1
        return self._pool[self._i % 8]
    def step(self, actions):
        actions = np.asarray(actions, np.float32).reshape(self.n_agents, -1)
        self.t += 1
        rew = np.tanh(actions[:, 0] * 0.01).astype(np.float32)
        done = self.t >= self.ep_len * 8
        self.t[done] = 0
        return self._obs(), rew, done.astype(np.float32), np.zeros(self.n_agents, np.float32), {"state": None}


with contextlib.redirect_stdout(sys.stderr):
    learner = Learner(FastVectorEnv, vector_env=True, n_proc=1, timestep_limit=10**9, exp_buffer_size=N_AGENTS * T,
                      ts_per_iteration=N_AGENTS * T, ppo_epochs=1, ppo_batch_size=N_AGENTS * T, ppo_minibatch_size=65536,
                      policy_layer_sizes=(256, 256, 256), critic_layer_sizes=(256, 256, 256), checkpoints_save_folder=None,
                      checkpoint_load_folder=None, save_every_ts=10**12, log_to_wandb=False, random_seed=3)
learner.ppo_learner.policy.noise_mode = os.environ.get("NOISE", "device")
sync = torch.cuda.synchronize
for it in range(3):
    sync(); t0 = time.perf_counter()
    exp, _, n, _ = learner.agent.collect_timesteps(N_AGENTS * T)
    sync(); t1 = time.perf_counter()
    learner.add_new_experience(exp)
    sync(); t2 = time.perf_counter()
    with contextlib.redirect_stdout(sys.stderr):
        learner.ppo_learner.learn(learner.experience_buffer)
    sync(); t3 = time.perf_counter()
    # the same experience handed over as host arrays (what BatchedAgentManager.collect_timesteps returns)
    host_exp = tuple(x.cpu().numpy()[:, :107] if x.dim() == 2 and x.shape[1] == 128 else x.cpu().numpy() for x in exp)
    sync(); t4 = time.perf_counter()
    learner.add_new_experience(host_exp)
    sync(); t5 = time.perf_counter()
    print(f"iteration {it}: collect {n} steps {(t1-t0)*1e3:7.1f} ms ({(t1-t0)/T*1e3:.2f} ms/step incl. the env) | add_new_experience "
          f"from the device rollout {(t2-t1)*1e3:6.1f} ms, from host arrays {(t5-t4)*1e3:6.1f} ms | PPO update (1 epoch) {(t3-t2)*1e3:6.1f} ms")
learner.agent.cleanup()
