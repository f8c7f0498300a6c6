"""bench.py's process_collect leg (the reference's default collection: worker processes through BatchedAgentManager and its wire format,
50,000 timesteps) with the learner-side loop in C++ (rlppo_collector_*, the default) and in Python (agent.native_collect = False),
alternating.  usage: python tools/process_collect_ab.py [n_proc ...]"""
import json, os, sys

if __name__ == "__main__":   # (the worker processes re-import the main module: nothing may run at import)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from rlgym_ppo_amd.batched_agents import BatchedAgentManager
    keys = ("n_proc", "steps_per_s", "seconds", "get_action_calls", "mean_obs_per_call", "us_per_get_action_median", "frac_of_wall_in_get_action",
            "transport", "spot_checked", "spot_check_mismatches", "collector", "add_new_experience_ms", "learn_ms", "iteration_steps_per_s", "error")
    init = BatchedAgentManager.__init__
    for n in ([int(x) for x in sys.argv[1:]] or [8, 32]):
        for rnd in range(2):
            for native in (True, False):
                def patched(self, *a, _native=native, **k):
                    init(self, *a, **k)
                    self.native_collect = _native
                BatchedAgentManager.__init__ = patched
                try:
                    r = bench.process_collect_leg(n)
                finally:
                    BatchedAgentManager.__init__ = init
                print("loop in %-6s %s" % ("C++" if native else "Python", json.dumps({k: r.get(k) for k in keys if r.get(k) is not None})), flush=True)
