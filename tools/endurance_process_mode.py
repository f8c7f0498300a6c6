"""The reference's training loop at its defaults -- Learner(n_proc 8, ts_per_iteration 50,000, ppo_batch_size = minibatch 50,000, 10 epochs,
buffer 150,000) -- run end to end for a fixed wall-clock budget on the synthetic two-agent environment of bench.py's process_collect leg:
worker processes -> wire format -> C++ collection loop -> policy.get_action -> add_new_experience (value pass, GAE, ring submit) ->
PPOLearner.learn.  Prints what every iteration took and checks that nothing drifts: parameters finite, every iteration collected what it was
asked for, the transport's counters, a checkpoint written and loaded back (the restored statistics are float64: the collection keeps its
C++ loop).  usage: python tools/endurance_process_mode.py [seconds] [n_proc]"""
import contextlib, os, shutil, sys, tempfile, time

if __name__ == "__main__":   # (the worker processes re-import the main module: nothing may run at import)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import numpy as np, torch
    import bench_process_env
    from rlgym_ppo_amd import Learner
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    n_proc = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    folder = tempfile.mkdtemp(prefix="rlppo_endurance_")
    rows = []
    try:
        for phase in ("fresh", "resumed"):
            with contextlib.redirect_stdout(sys.stderr):
                learner = Learner(bench_process_env.make_env, n_proc=n_proc, min_inference_size=80, timestep_limit=10**12, exp_buffer_size=150_000,
                                  ts_per_iteration=50_000, ppo_epochs=10, ppo_batch_size=50_000, ppo_minibatch_size=50_000,
                                  checkpoints_save_folder=folder, add_unix_timestamp=False, checkpoint_load_folder="latest" if phase == "resumed" else None,
                                  save_every_ts=10**12, log_to_wandb=False, random_seed=7)
            try:
                if phase == "resumed":
                    assert learner.agent.obs_stats.running_mean.dtype == np.float64 and learner.agent.cumulative_timesteps > 0
                t0, it = time.perf_counter(), 0
                while time.perf_counter() - t0 < budget / 2:
                    t = time.perf_counter()
                    exp, _, n, secs = learner.agent.collect_timesteps(50_000)
                    t1 = time.perf_counter()
                    learner.add_new_experience(exp)
                    torch.cuda.synchronize()
                    t2 = time.perf_counter()
                    rep = learner.ppo_learner.learn(learner.experience_buffer)
                    t3 = time.perf_counter()
                    assert n >= 50_000 and len(exp[0]) > 0 and learner.agent._native is not None
                    assert all(np.isfinite(v) for k, v in rep.items() if isinstance(v, float)), rep
                    rows.append((phase, it, n / (t1 - t), (t2 - t1) * 1e3, (t3 - t2) * 1e3, n / (t3 - t), rep["Policy Entropy"], rep["Mean KL Divergence"]))
                    it += 1
                for m in (learner.ppo_learner.policy, learner.ppo_learner.value_net):
                    assert all(torch.isfinite(p).all() for p in m.parameters())
                g = list(learner.ppo_learner.policy._graphs.values())
                tr = {k: sum(getattr(x, k) for x in g) for k in ("calls", "polled", "poll_timeouts", "late_retries", "stale_relaunches")}
                print("%s: %d iterations, %d timesteps, %d optimiser steps; average reward %.4f; observation statistics %s, count %d; transport %s"
                      % (phase, it, learner.agent.cumulative_timesteps, learner.ppo_learner.cumulative_model_updates, learner.agent.average_reward,
                         learner.agent.obs_stats.running_mean.dtype, learner.agent.obs_stats.count, tr), flush=True)
                if phase == "fresh":
                    with contextlib.redirect_stdout(sys.stderr):
                        learner.save(learner.agent.cumulative_timesteps)
            finally:
                with contextlib.redirect_stdout(sys.stderr):
                    learner.cleanup()
        a = np.array([r[2:] for r in rows], dtype=np.float64)
        print("per iteration (median / min / max over %d): collect %.0f / %.0f / %.0f k steps/s; add_new_experience %.2f / %.2f / %.2f ms; learn (10 epochs x 3 batches) "
              "%.2f / %.2f / %.2f ms; whole iteration %.0f / %.0f / %.0f k steps/s" % (
                  len(rows), np.median(a[:, 0]) / 1e3, a[:, 0].min() / 1e3, a[:, 0].max() / 1e3, np.median(a[:, 1]), a[:, 1].min(), a[:, 1].max(),
                  np.median(a[:, 2]), a[:, 2].min(), a[:, 2].max(), np.median(a[:, 3]) / 1e3, a[:, 3].min() / 1e3, a[:, 3].max() / 1e3))
        print("policy entropy first -> last: %.4f -> %.4f; KL of the last iteration %.2e" % (rows[0][6], rows[-1][6], rows[-1][7]))
    finally:
        shutil.rmtree(folder, ignore_errors=True)
