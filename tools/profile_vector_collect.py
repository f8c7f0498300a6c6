"""cProfile of the device-resident collection of bench.py's iteration leg (VectorAgentManager.collect_timesteps, 4096 agents x 128 steps,
device noise so that the host noise pipeline is out of the picture).  usage: python tools/profile_vector_collect.py"""
import cProfile, contextlib, io, os, pstats, sys

if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import bench
    from rlgym_ppo_amd import Learner
    with contextlib.redirect_stdout(sys.stderr):
        learner = Learner(bench.BenchVectorEnv, vector_env=True, n_proc=1, timestep_limit=10**9, exp_buffer_size=bench.N_SAMPLES,
                          ts_per_iteration=bench.N_SAMPLES, ppo_epochs=10, ppo_batch_size=bench.BATCH, ppo_minibatch_size=bench.MINIBATCH,
                          policy_layer_sizes=bench.HID, critic_layer_sizes=bench.HID, checkpoints_save_folder=None,
                          checkpoint_load_folder=None, save_every_ts=10**12, log_to_wandb=False, random_seed=123)
    try:
        for mode in ("device", "host"):
            learner.ppo_learner.policy.noise_mode = mode
            for _ in range(2):
                learner.agent.collect_timesteps(bench.N_SAMPLES)
            torch.cuda.synchronize()
            pr = cProfile.Profile()
            pr.enable()
            learner.agent.collect_timesteps(bench.N_SAMPLES)
            torch.cuda.synchronize()
            pr.disable()
            s = io.StringIO()
            pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
            print("---- noise_mode", mode)
            print("\n".join(l[:150] for l in s.getvalue().splitlines() if l.strip()))
    finally:
        learner.agent.cleanup()
