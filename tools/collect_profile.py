"""Host-side profile of VectorAgentManager.collect_timesteps at the configs[1] scale (4096 agents x 128 steps).
usage: python tools/collect_profile.py [device|host]   (noise mode)"""
import contextlib, cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rlgym_ppo_amd import Learner

with contextlib.redirect_stdout(sys.stderr):
    learner = Learner(bench.BenchVectorEnv, vector_env=True, n_proc=1, timestep_limit=10**9, exp_buffer_size=bench.N_SAMPLES,
                      ts_per_iteration=bench.N_SAMPLES, ppo_epochs=1, ppo_batch_size=bench.BATCH, ppo_minibatch_size=bench.MINIBATCH,
                      policy_layer_sizes=bench.HID, critic_layer_sizes=bench.HID, checkpoints_save_folder=None,
                      checkpoint_load_folder=None, save_every_ts=10**12, log_to_wandb=False, random_seed=123)
if len(sys.argv) > 1:
    learner.ppo_learner.policy.noise_mode = sys.argv[1]
for _ in range(2):
    t = time.perf_counter()
    learner.agent.collect_timesteps(bench.N_SAMPLES)
    torch.cuda.synchronize()
    print("collect: %.1f ms (%.3f ms per env step)" % ((time.perf_counter() - t) * 1e3, (time.perf_counter() - t) * 1e3 / bench.N_STEPS))
pr = cProfile.Profile()
pr.enable()
learner.agent.collect_timesteps(bench.N_SAMPLES)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr, stream=sys.stdout).sort_stats("tottime").print_stats(18)
learner.agent.cleanup()
