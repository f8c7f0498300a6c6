"""A/B of the GAE scan (BASELINE configs[2]: 8192 x 256 steps): compile-time steps per thread (RLPPO_LIB picks the build) x
grid policy (rlppo_dbg_set(22): clamp to the resident capacity | one workgroup per chunk).  One subprocess per build."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    import bench
    from rlgym_ppo_amd import _native as N
    L = N.lib()
    rs = np.random.RandomState(0)
    n = 8192 * 256
    R, V = torch.as_tensor(rs.randn(n).astype(np.float32)).cuda(), torch.as_tensor(rs.randn(n + 1).astype(np.float32)).cuda()
    D = torch.as_tensor((rs.rand(n) < 0.005).astype(np.float32)).cuda()
    T = torch.zeros(n, device="cuda")
    T[255::256] = 1
    vt, adv, ret = (torch.empty(n, device="cuda") for _ in range(3))
    ws = torch.zeros(int(L.rlppo_gae_workspace_bytes(n)), dtype=torch.uint8, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    fn = lambda: N.check(L.rlppo_gae(st, P(R), P(D), P(T), P(V), n, 0.99, 0.95, float(np.float32(1.7)), P(vt), P(adv), P(ret), P(ws), ws.numel()))
    fn()
    ref = adv.clone()
    bench.time_region(fn, 1, warm_s=0.3)
    res = {0: [], 1: []}
    for _ in range(7):
        for over in (0, 1):
            N.check(L.rlppo_dbg_set(22, over))
            res[over].append(bench.time_region(fn, 20, warm=2))
            assert torch.equal(adv, ref)
    if not os.environ.get("RLPPO_LIB"):  # once: what the memory system allows for a launch of this size and byte count
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import _diag
        o = [torch.empty(n, device="cuda") for _ in range(3)]
        floor = lambda shape: (lambda: _diag.check(_diag.DL.rlppo_dbg_stream_floor(st, P(R), P(D), P(T), P(V), P(o[0]), P(o[1]), P(o[2]), n, shape)))
        src, dst = torch.empty(28 * n // 8, device="cuda"), torch.empty(28 * n // 8, device="cuda")
        bench.time_region(floor(0), 1, warm_s=0.3)
        fl = [bench.time_region(floor(0), 20, warm=2) for _ in range(7)]
        fl1 = [bench.time_region(floor(1), 20, warm=2) for _ in range(7)]
        cp = [bench.time_region(lambda: dst.copy_(src), 20, warm=2) for _ in range(7)]
        for name, ts in (("streaming floor (same streams, 8 consecutive steps per thread)", fl),
                         ("streaming floor, lane-contiguous float4 (1 KiB per wave-instruction)", fl1),
                         ("device copy of 29.4 MB (same bytes moved)", cp)):
            us = float(np.median(ts)) * 1e3
            print(f"{name:>72s}: {us:6.2f} us  {28 * n / us / 1e6:6.2f} TB/s  ({28 * n / us / 8e6:.3f} of 8 TB/s)")
    for over in (0, 1):
        us = float(np.median(res[over])) * 1e3
        print(f"{os.environ.get('RLPPO_LIB', 'default build'):>40s}  oversubscribe={over}: {us:6.2f} us  {28 * n / us / 1e6:6.2f} TB/s  ({28 * n / us / 8e6:.3f} of 8 TB/s)")
else:
    for lib in sys.argv[1:] or [""]:
        env = dict(os.environ)
        if lib:
            env["RLPPO_LIB"] = os.path.join(ROOT, "rlgym_ppo_amd", lib)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
