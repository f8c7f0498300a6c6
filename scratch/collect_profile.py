import os, sys, time, contextlib, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
sys.argv = ["x"]
import importlib.util
spec = importlib.util.spec_from_file_location("ib", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "iteration_bench.py"))
src = open(spec.origin).read().split("sync = torch.cuda.synchronize")[0]
exec(compile(src, "ib", "exec"))
learner.ppo_learner.policy.noise_mode = "device"
learner.agent.collect_timesteps(N_AGENTS * 16)
pr = cProfile.Profile(); pr.enable()
learner.agent.collect_timesteps(N_AGENTS * 64)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
