"""cProfile of the learner-side of the reference's process-mode collection (bench.py's process_collect leg): where BatchedAgentManager
spends its wall clock.  usage: python tools/profile_process_collect.py [n_proc]"""
import cProfile, io, os, pstats, sys

if __name__ == "__main__":   # (the worker processes re-import the main module: nothing may run at import)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    n_proc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    pr = cProfile.Profile()
    pr.enable()
    r = bench.process_collect_leg(n_proc)
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
    print("\n".join(l[:160] for l in s.getvalue().splitlines()))
    print(r)
