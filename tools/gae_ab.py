"""A/B of the GAE scan (BASELINE configs[2]: 8192 x 256 steps): compile-time variants (RLPPO_LIB picks the build) x grid policy
(rlppo_dbg_set(22): clamp to the resident capacity | one workgroup per chunk), each COLD (rotating over bench.GAE_SETS buffer sets:
an HBM measurement) and HOT (one set re-scanned in the Infinity Cache), beside the streaming floors and a device copy of the same
bytes measured the same two ways.  One subprocess per build."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    import bench
    from rlgym_ppo_amd import _native as N
    L = N.lib()
    S = bench.GAE_SETS
    fns, sets, n = bench.gae_sets(S)       # S independent input + output sets: the rotation makes every launch cold (bench.gae_bench)
    for fn in fns:
        fn()
    ref = sets[0][1][1].clone()
    bench.time_region(fns[0], 1, warm_s=0.3)
    med = lambda ts: float(np.median(ts)) * 1e3
    line = lambda name, cold, hot: print(f"{name:>110s}: cold {cold:6.2f} us {28 * n / cold / 1e6:5.2f} TB/s ({28 * n / cold / 8e6:.3f} of 8 TB/s) | "
                                         f"hot {hot:6.2f} us ({28 * n / hot / 8e6:.3f})", flush=True)
    FORMS = {1: "single launch"}
    res = {(a, o, k): [] for a in FORMS for o in (0, 1) for k in "ch"}
    for _ in range(7):
        for algo in FORMS:
            N.check(L.rlppo_dbg_set(1, algo))
            for over in (0, 1):
                N.check(L.rlppo_dbg_set(22, over))
                res[(algo, over, "h")].append(bench.time_region(fns[0], 20, warm=2))
                res[(algo, over, "c")].append(bench.time_rotating(fns, 3))
                assert all(torch.equal(o[1], ref) for _, o in sets)
    N.check(L.rlppo_dbg_set(22, 0))
    N.check(L.rlppo_dbg_set(1, 1))
    if not os.environ.get("RLPPO_LIB"):  # once: what the memory system allows for a launch of this size and byte count
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import _diag
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        def floor(shape, i):
            (R, D, T, V), o = sets[i]
            return lambda: _diag.check(_diag.DL.rlppo_dbg_stream_floor(st, P(R), P(D), P(T), P(V), P(o[0]), P(o[1]), P(o[2]), n, shape))
        pairs = [(torch.empty(28 * n // 8, device="cuda"), torch.empty(28 * n // 8, device="cuda")) for _ in range(S)]
        cps = [lambda a=a, b=b: b.copy_(a) for a, b in pairs]
        for name, fs in (("streaming floor (same 7 streams, 8 consecutive steps per thread, elementwise map)", [floor(0, i) for i in range(S)]),
                         ("streaming floor, lane-contiguous float4 (1 KiB per wave-instruction)", [floor(1, i) for i in range(S)]),
                         ("streaming floor, lane-contiguous LOADS, 8-consecutive stores", [floor(2, i) for i in range(S)]),
                         ("streaming floor, 8-consecutive loads, lane-contiguous STORES", [floor(3, i) for i in range(S)]),
                         ("device copy of the same bytes (29.4 MB read + 29.4 MB written)", cps)):
            bench.time_region(fs[0], 1, warm_s=0.2)
            line(name, med([bench.time_rotating(fs, 3) for _ in range(7)]), med([bench.time_region(fs[0], 20, warm=2) for _ in range(7)]))
    for algo in FORMS:
        for over in (0, 1):
            line(f"{os.path.basename(os.environ.get('RLPPO_LIB', 'default build'))}: {FORMS[algo]}, oversubscribe={over}",
                 med(res[(algo, over, "c")]), med(res[(algo, over, "h")]))
else:
    for lib in sys.argv[1:] or [""]:
        env = dict(os.environ)
        if lib:
            env["RLPPO_LIB"] = os.path.join(ROOT, "rlgym_ppo_amd", lib)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
