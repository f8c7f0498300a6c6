#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json): PPO-update samples/s.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload = BASELINE.json configs[1] ("1xMI355X: 4096 parallel agents, obs=107 f32, |A|=90 discrete, 256x3 MLP,
512k-sample buffer"): a 524,288-sample device-resident ExperienceBuffer (4096 agents x 128 steps), ppo_batch_size
524,288, ppo_minibatch_size 65,536, lr 3e-4, clip 0.2, ent 0.005, synthetic data, random-init
weights.  One *step* = one PPOLearner.learn() call = `--epochs` (default 10, the reference's ppo_epochs default,
learner.py:51) shuffled passes over the whole buffer, i.e. epochs x 524,288 samples through forward + loss + backward
+ clip + Adam, including the host-side legacy-MT19937 shuffle and the index upload.  value = samples / wall time,
whole job (all ranks).  With N ranks the 8 minibatch slices of every batch are dealt to the ranks in contiguous blocks (a rank
evaluates its consecutive slices in one fused pass: the minibatches of a batch are independent and their gradients are summed
before the one optimiser step, ppo_learner.py:134-193) and one RCCL all-reduce of the flat gradient arena precedes the optimiser
step (total work fixed: strong scaling).

Extra objects on the same JSON line (N=1 only): `roofline` for the dominant kernel of the timed region, `gae` =
the GAE scan (BASELINE configs[2]: 8192 x 256, HBM-bound) with its own roofline, `rollout` = policy inference
obs/s at 4096 x 107, `cpu_baseline` = the CPU oracle (a port of the reference's op sequence; kind "port") timed on
this box's host cores on a bounded sample.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_F32_PEAK_TF = 157.3    # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak
OBS, ACT, HID = 107, 90, (256, 256, 256)
N_AGENTS, N_STEPS = 4096, 128
N_SAMPLES = N_AGENTS * N_STEPS  # 524,288
BATCH, MINIBATCH = 524288, 65536
FLOP_PER_SAMPLE = 1_931_776     # fwd + bwd of both nets, cfg2


def log(*a):
    print(*a, file=sys.stderr, flush=True)


CFG5 = dict(obs=231, act=8, hid=(512, 512, 512, 512), flop_per_sample=10_435_584)  # BASELINE configs[4], SURVEY 8(d)


def build_workload(device, seed=123, config="cfg2"):
    from rlgym_ppo_amd.ppo import ExperienceBuffer, PPOLearner
    torch.manual_seed(seed)
    np.random.seed(seed)
    gauss = config == "cfg5"
    obs_d, hid = (CFG5["obs"], CFG5["hid"]) if gauss else (OBS, HID)
    learner = PPOLearner(obs_d, CFG5["act"] if gauss else ACT, 2 if gauss else 0, hid, hid, (0.1, 1.0), BATCH, 10, 3e-4, 3e-4, 0.2,
                         0.005, MINIBATCH, device)
    g = torch.Generator(device=device).manual_seed(seed)
    states = torch.randn(N_SAMPLES, obs_d, device=device, generator=g).clamp_(-5, 5)
    acts, logps = [], []
    for s in range(0, N_SAMPLES, 65536):  # actions sampled by the policy itself at init weights -> ratio ~ 1 at step 0
        if gauss:
            noise = torch.empty(65536, CFG5["act"], device=device).normal_(0, 1, generator=g)
        else:
            noise = torch.empty(65536, ACT, device=device).exponential_(1, generator=g)
        a, lp = learner.policy.get_action(states[s:s + 65536], noise=noise)
        acts.append(a)
        logps.append(lp)
    actions = torch.cat(acts).float()
    log_probs = torch.cat(logps)
    adv = torch.randn(N_SAMPLES, device=device, generator=g)
    tgt = torch.randn(N_SAMPLES, device=device, generator=g)
    z = torch.zeros(N_SAMPLES, device=device)
    buf = ExperienceBuffer(N_SAMPLES, seed, "cpu")
    buf.submit_experience(states, actions, log_probs, z, states[:1].expand(N_SAMPLES, obs_d), z, z, tgt, adv)
    return learner, buf


# ------------------------------------------------------------------------------------------- kernel timing
def time_region(fn, reps, warm=3, warm_s=0.0):
    """Average device time of fn() in ms, HIP events on torch's current stream (the stream librlppo launches on).
    warm_s > 0 first keeps the GPU busy with fn() for that long: the shader clock needs a few hundred ms of load to
    ramp from ~2.06 to ~2.35 GHz (tools/clock_under_load.py), and the timed region of the bench runs at the ramped clock."""
    for _ in range(warm):
        fn()
    if warm_s > 0:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < warm_s:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def kernel_breakdown(learner, only=None):
    """Times every GEMM launch shape of one pass of the update through the diagnostic entry points, at the row count the
    update really launches (learner._fused_rows: at one GPU the 8 minibatches of a batch are evaluated in ONE pass of
    524,288 rows, PPOLearner.max_fused_minibatches; with 8 ranks a pass is one 65,536-row minibatch), and returns
    (rows, dominant) where rows = [(name, launches per pass, ms per launch, flop per launch)]."""
    from rlgym_ppo_amd import _native as N
    L = N.lib()
    dev = learner._dev
    M = int(getattr(learner, "_fused_rows", MINIBATCH))
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    K0 = int(L.rlppo_padded_width(OBS))          # 112: the first layer's padded contraction (was 128 until round 3)
    A0 = torch.randn(M, K0, device=dev)
    A0[:, OBS:] = 0
    A256 = torch.randn(M, 256, device=dev)
    A256b = torch.randn(M, 256, device=dev)  # second operand of dW / mask of dX: distinct memory, as in the update
    A96 = torch.randn(M, 96, device=dev)
    W = torch.randn(256, 256, device=dev) * 0.05
    bias = torch.zeros(256, device=dev)
    C256 = torch.empty(M, 256, device=dev)
    C96 = torch.empty(M, 96, device=dev)
    dW = torch.zeros(256 * 256, device=dev)
    db = torch.zeros(256, device=dev)

    def nt(A, lda, ldb, C, ldc, n, k, epi, mask=None):
        return lambda: N.check(L.rlppo_dbg_gemm_nt(st(), P(A), lda, P(W), ldb, P(bias), P(mask) if mask is not None else None, n,
                                                    P(C), ldc, M, n, k, epi))

    tn_ws = torch.empty(max(max(int(L.rlppo_dbg_gemm_tn_workspace_bytes(o, i, M)) for o, i in ((256, 256), (256, 107), (90, 256))), 1),
                        dtype=torch.uint8, device=dev)

    # the forms the update launches for 128-multiple widths: the hidden-layer forward also writes the ReLU bitmask, the masked
    # dX product reads it (csrc/gemm.hip); the bitmask buffer here is written by the forward launches timed first
    bits = torch.zeros(max(int(L.rlppo_dbg_gemm_nt_bits_bytes(M, 256)), 8), dtype=torch.uint8, device=dev)

    def ntb(A, lda, C, ldc, n, k, epi):
        return lambda: N.check(L.rlppo_dbg_gemm_nt_bits(st(), P(A), lda, P(W), lda, P(bias) if epi == 1 else None, P(C), ldc, M, n, k,
                                                        epi, P(bits)))

    def tn(dY, ny, X, kx, out, in_):  # partial tiles + reduction kernel (both timed)
        return lambda: N.check(L.rlppo_dbg_gemm_tn(st(), P(dY), ny, ny, P(X), kx, kx, P(dW), P(db), out, in_, M, P(tn_ws),
                                                    tn_ws.numel()))

    # [r5] the update launches every weight-gradient product of a pass as ONE grouped launch + ONE reduction (csrc/gemm.hip,
    # rlppo_dbg_set(37)): 2 first-layer, 4 hidden, 1 policy-head product (the critic's one-output head is a matrix-vector kernel)
    gshapes = [(256, OBS, A256, 256, A0, K0), (256, 256, A256, 256, A256b, 256), (256, 256, A256b, 256, A256, 256), (ACT, 256, A96, 96, A256, 256),
               (256, OBS, A256b, 256, A0, K0), (256, 256, A256, 256, A256b, 256), (256, 256, A256b, 256, A256, 256)]
    gprods = (N.TnProduct * len(gshapes))()
    gkeep = []
    for q, (out, in_, dY, ny, X, kx) in zip(gprods, gshapes):
        gw, gb = torch.zeros(out * in_, device=dev), torch.zeros(out, device=dev)
        gkeep.append((gw, gb))
        q.dY, q.ldy, q.ny_valid, q.X, q.ldx, q.kx_valid = dY.data_ptr(), ny, ny, X.data_ptr(), kx, kx
        q.dW, q.db, q.out, q.in_, q.rowtab, q.src_rows = gw.data_ptr(), gb.data_ptr(), out, in_, None, 0
    g_ws = torch.empty(max(int(L.rlppo_dbg_gemm_tn_group_workspace_bytes(gprods, len(gshapes), M)), 1), dtype=torch.uint8, device=dev)
    tn_group = lambda: N.check(L.rlppo_dbg_gemm_tn_group(st(), gprods, len(gshapes), M, P(g_ws), g_ws.numel()))
    g_exec = sum(f2 for f2 in (2 * M * 256 * K0, 2 * M * 256 * 256, 2 * M * 256 * 256, 2 * M * 96 * 256, 2 * M * 256 * K0, 2 * M * 256 * 256, 2 * M * 256 * 256))
    g_alg = sum(2 * M * o * i for o, i, *_ in gshapes)

    # (name, launches per pass, launch, EXECUTED flop = the padded tile shape the kernel multiplies, ALGORITHMIC flop = the logical
    # layer shape: K = 107 observations, N = 90 actions).  The rates printed as `tflops` / `frac` are the algorithmic ones.
    f = lambda n, k: 2 * M * n * k
    shapes = [
        ("fwd L0 %d->256" % K0, 2, ntb(A0, K0, C256, 256, 256, K0, 1), f(256, K0), f(256, OBS)),
        ("fwd hidden 256->256", 4, ntb(A256, 256, C256, 256, 256, 256, 1), f(256, 256), f(256, 256)),
        ("fwd head 256->96", 1, nt(A256, 256, 256, C96, 96, 96, 256, 0), f(96, 256), f(ACT, 256)),
        ("dX hidden 256->256", 4, ntb(A256, 256, C256, 256, 256, 256, 3), f(256, 256), f(256, 256)),
        ("dX head 96->256", 1, ntb(A96, 96, C256, 256, 256, 96, 3), f(256, 96), f(256, ACT)),
        ("dW all 7 products, grouped + reduce", 1, tn_group, g_exec, g_alg),
        # the per-layer form (one launch + reduction per product; what rounds 1-4 ran, rlppo_dbg_set(37, 0)): n = 0, not part of a pass
        ("dW hidden 256x256 (per-layer form)", 0, tn(A256, 256, A256b, 256, 256, 256), f(256, 256), f(256, 256)),
        ("dW L0 256x107 (per-layer form)", 0, tn(A256, 256, A0, K0, 256, 107), f(256, K0), f(256, OBS)),
        ("dW head 90x256 (per-layer form)", 0, tn(A96, 96, A256, 256, 90, 256), f(96, 256), f(ACT, 256)),
    ]
    rows = []
    for name, count, fn, flop_exec, flop in shapes:
        if only is not None and not name.startswith(only):
            continue
        ms = time_region(fn, 20, warm_s=0.3)
        # compact keys (the driver keeps ~12 KB of the line): n = launches per pass, ms = per launch, tflops / frac on ALGORITHMIC flop
        # (K = 107, N = 90), exec_tflops on the padded tile shape the kernel multiplies
        rows.append(dict(kernel=name, n=count, ms=round(ms, 4), gflop=round(flop / 1e9, 2), tflops=round(flop / ms / 1e9, 1),
                         frac=round(flop / ms / 1e9 / MFMA_F32_PEAK_TF, 3), exec_tflops=round(flop_exec / ms / 1e9, 1)))
    dominant = max(rows, key=lambda r: r["n"] * r["ms"])
    return rows, dominant, M


# tools/pmc_traffic.py output of the committed PMC passes (tools/round_profile.sh), newest round first.  Every `traffic` on the JSON
# line is READ FROM THESE FILES (builder-run rocprofv3 --pmc passes: bench.py cannot profile itself) and carries a `traffic_source`.
# [r6] Each file is stamped with rlppo_build_id() of the library it was measured on; a number is replayed only while that is the
# library this process has loaded -- after a kernel change the line says `traffic: null, stale` until the pass is regenerated.
PMC_NOTE = " (builder-run rocprofv3 --pmc passes, replayed)"


def build_id():
    from rlgym_ppo_amd import _native as N
    return N.lib().rlppo_build_id().decode()


def replay_traffic(stem, key):
    """(HBM bytes per launch, source) of `key` from the newest profiles/<round>_<stem>.json whose build id is this library's."""
    mine, seen = build_id(), []
    for tag in ("r06", "r05", "r04", "r03", "r02"):
        name = tag + "_" + stem + ".json"
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        try:
            t = json.load(open(path))
            theirs = t.get("_build_id")
            if theirs != mine:
                seen.append("%s (build %s)" % (name, theirs or "unstamped"))
                continue
            return round(t[key]["hbm_bytes"]), "build " + mine + ": profiles/" + name + PMC_NOTE
        except Exception:
            continue
    if seen:
        return None, "stale: " + seen[0] + " was not measured on this library (build " + mine + ")"
    return None, "no committed PMC pass for this kernel"


def pmc_traffic_for(kernel_label):
    """HBM bytes per launch of the dominant kernel, from the committed PMC passes (bench.py cannot run rocprofv3 on itself)."""
    # tools/prof_kernels.py launches the same shapes as kernel_breakdown (M = 524,288 rows); tools/pmc_summary.py tells the
    # three dW shapes (same kernel, same grid) apart by their position in the launch cycle
    key = {"dW all 7 products, grouped + reduce": "rlppo::gemm_tn_group_kernel",
           "dW hidden 256x256": "rlppo::gemm_tn_dma_kernel<32, 4, 4, 2, false>",
           "dW L0 256x107": "rlppo::gemm_tn_dma_kernel<32, 2, 7, 4, false>",
           "dW head 90x256": "rlppo::gemm_tn_dma_kernel<32, 3, 4, 2, false>",
           "fwd hidden 256->256": "rlppo::gemm_nt_dma_kernel<8, 1, 16, true, false> {fwd hidden 256->256}",
           "fwd L0 112->256": "rlppo::gemm_nt_dma_kernel<8, 1, 16, true, false> {fwd L0 112->256}",
           "fwd head 256->96": "rlppo::gemm_nt_dma_kernel<6, 0, 16, false, false>",
           "dX hidden 256->256": "rlppo::gemm_nt_dma_kernel<8, 3, 16, true, false> {dX hidden 256->256}",
           "dX head 96->256": "rlppo::gemm_nt_dma_kernel<8, 3, 16, true, false> {dX head 96->256}"}.get(kernel_label)
    if key is None:
        return None, "no committed PMC pass for this kernel"
    return replay_traffic("traffic", key)


def non_gemm_tail():
    """The update's non-GEMM kernels (loss, the critic head's matrix-vector kernels, reductions, optimiser tail, gather) per epoch,
    from the newest committed single-stream kernel trace of `bench.py --steps 2 --warmup 1 --no-extras` (3 learn() x 10 epochs = 30
    passes; tools/round_profile.sh): {kernel: us per epoch}.  Like `traffic`, read from profiles/ -- bench.py cannot trace itself."""
    import csv
    for tag in ("r06", "r05", "r04", "r03"):
        path = os.path.join(ROOT, "profiles", tag + "_bench_kernel_stats_single_stream.csv")
        if not os.path.exists(path):
            continue
        try:
            rows, tail = list(csv.DictReader(open(path))), {}
            adam = [r for r in rows if "adam_fused_kernel" in r["Name"] or "adam_pack2_kernel" in r["Name"]]
            passes = int(adam[0]["Calls"]) if adam else 30
            for r in rows:
                name = r["Name"].replace("void ", "")
                if not name.startswith("rlppo::") or "gemm_nt" in name or "gemm_tn" in name or int(r["Calls"]) < passes:
                    continue
                short = name.split("(")[0].replace("rlppo::", "")
                tail[short] = round(tail.get(short, 0.0) + float(r["TotalDurationNs"]) / passes / 1e3, 1)
            return dict(us_per_epoch=tail, total_us_per_epoch=round(sum(tail.values()), 1),
                        source="profiles/" + tag + "_bench_kernel_stats_single_stream.csv (builder-run rocprofv3 --kernel-trace --stats, single stream)")
        except Exception:
            continue
    return dict(us_per_epoch=None, source="no committed kernel trace")


def gae_traffic():
    """HBM bytes per scan from the committed PMC passes over tools/prof_gae.py (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE),
    newest round first; the passes rotate over GAE_SETS buffer sets exactly as the timed region below does."""
    return replay_traffic("gae_traffic", "rlppo::gae_lookback_kernel<false>")


GAE_SETS = 10  # buffer sets the cold measurement rotates over: 10 x 58.7 MB of inputs + outputs = 587 MB between two uses of a
               # line, against 256 MiB of Infinity Cache (MI355X_MICROARCH.md) + 8 x 4 MiB of L2: every timed scan streams from HBM


def gae_inputs(seed=0, n_seg=8192, seg=256):
    """BASELINE configs[2] / SURVEY 8(d): 8192 segments x 256 steps, every segment ends in done or truncated (p = 0.5 each), iid
    mid-segment dones p = 0.005, seed 0."""
    rs = np.random.RandomState(seed)
    n = n_seg * seg
    rews = rs.randn(n).astype(np.float32)
    values = rs.randn(n + 1).astype(np.float32)
    dones = (rs.rand(n) < 0.005).astype(np.float32)
    trunc = np.zeros(n, np.float32)
    ends = np.arange(seg - 1, n, seg)
    is_done = rs.rand(n_seg) < 0.5
    dones[ends[is_done]] = 1
    dones[ends[~is_done]] = 0
    trunc[ends[~is_done]] = 1
    return rews, dones, trunc, values


def gae_sets(n_sets=GAE_SETS, host=None):
    """n_sets independent copies of the configs[2] inputs AND outputs in HBM (distinct allocations) + one launch closure per set.
    Returns (launchers, sets, n): launchers[i]() enqueues one rlppo_gae scan over set i on torch's current stream."""
    from rlgym_ppo_amd import _native as N
    L = N.lib()
    rews, dones, trunc, values = host if host is not None else gae_inputs()
    n = rews.shape[0]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws = torch.zeros(int(L.rlppo_gae_workspace_bytes(n)), dtype=torch.uint8, device="cuda")
    sets, launchers = [], []
    for _ in range(n_sets):
        ins = [torch.as_tensor(x).cuda() for x in (rews, dones, trunc, values)]
        outs = [torch.empty(n, device="cuda") for _ in range(3)]
        cargs = [ctypes.c_void_p(t.data_ptr()) for t in ins] + [n, 0.99, 0.95, float(np.float32(1.7))] + \
                [ctypes.c_void_p(t.data_ptr()) for t in outs + [ws]] + [ws.numel()]
        sets.append((ins, outs))
        launchers.append(lambda cargs=cargs: N.check(L.rlppo_gae(st, *cargs)))
    return launchers, sets, n


def time_rotating(fns, cycles, warm_cycles=1):
    """Average device time per call in ms of the launches fns[0], fns[1], ... taken round-robin, `cycles` times through the
    list, HIP events on torch's current stream.  With every fn working on its own buffer set, a set is touched once per cycle."""
    for _ in range(warm_cycles):
        for fn in fns:
            fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(cycles):
        for fn in fns:
            fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / (cycles * len(fns))


def gae_bench():
    """BASELINE configs[2]: 8192 trajectories x 256 steps, gamma .99 lambda .95, return_std 1.7, fp32, seed 0.

    COLD (the roofline number): the timed scans rotate over GAE_SETS independent input + output sets, so that no scan finds its
    lines in the Infinity Cache or in L2 -- bytes / time is a rate against HBM.  HOT (`hot_frac`, rounds 1-3's number): the same
    set scanned back to back, its 58.7 MB resident in the 256 MiB memory-side cache.  The device-copy probes beside them move
    the same number of bytes (29.4 MB read + 29.4 MB written) cold and hot: the HBM floor of a launch of this size."""
    from rlgym_ppo_amd import _native as N
    from rlgym_ppo_amd.util import torch_functions
    host = gae_inputs()
    fns, sets, n = gae_sets(GAE_SETS, host)
    L = N.lib()
    for fn in fns:
        fn()
    (R, D, T, V), outs0 = sets[0]
    chk = torch_functions.gae_device(R, D, T, V, 0.99, 0.95, 1.7)
    for _, outs in sets:  # every set holds the same inputs: every scan of the rotation is checked against the product entry point
        assert all(torch.equal(a, b) for a, b in zip(outs, chk))
    time_region(fns[0], 1, warm_s=0.3)  # clock ramp
    # single launch (default) against the two-launch form, hot and cold, interleaved in one process (cdna_hip_programming.md rule 24)
    hot, cold = {0: [], 1: []}, {0: [], 1: []}
    for _ in range(5):
        for algo in (0, 1):
            N.check(L.rlppo_dbg_set(1, algo))
            hot[algo].append(time_region(fns[0], 20, warm=2))
            cold[algo].append(time_rotating(fns, 3))
    N.check(L.rlppo_dbg_set(1, 1))
    ms_hot, ms_cold = float(np.median(hot[1])), float(np.median(cold[1]))
    # the same bytes through a device copy, cold (rotating) and hot
    nb = 28 * n // 2
    pairs = [(torch.empty(nb, dtype=torch.uint8, device="cuda"), torch.empty(nb, dtype=torch.uint8, device="cuda")) for _ in range(GAE_SETS)]
    cps = [lambda a=a, b=b: b.copy_(a) for a, b in pairs]
    cp_cold = float(np.median([time_rotating(cps, 3) for _ in range(5)]))
    cp_hot = float(np.median([time_region(cps[0], 20, warm=2) for _ in range(5)]))
    del pairs, cps
    alg_bytes = 28 * n
    traffic, traffic_src = gae_traffic()
    frac = lambda ms: round(alg_bytes / ms / 1e6 / HBM_PEAK_GBS, 4)
    log("gae: cold %.2f us (%.3f of 8 TB/s), hot %.2f us (%.3f); two-launch form cold %.2f / hot %.2f us; device copy of the same "
        "bytes cold %.2f us (%.3f) / hot %.2f us (%.3f).  cold = %d rotating input+output sets (%.0f MB between two uses of a line), "
        "HIP events around 3 x %d back-to-back scans, median of 5 rounds; hot = one set scanned 20 x back to back"
        % (ms_cold * 1e3, frac(ms_cold), ms_hot * 1e3, frac(ms_hot), float(np.median(cold[0])) * 1e3, float(np.median(hot[0])) * 1e3,
           cp_cold * 1e3, frac(cp_cold), cp_hot * 1e3, frac(cp_hot), GAE_SETS, GAE_SETS * alg_bytes / 1e6, GAE_SETS))
    out = dict(workload="configs[2]: 8192 x 256 steps f32", steps=n, us_per_scan=round(ms_cold * 1e3, 2), us_per_scan_hot=round(ms_hot * 1e3, 2),
               us_two_launch_form=round(float(np.median(cold[0])) * 1e3, 2), rotating_sets=GAE_SETS,
               us_copy_same_bytes=round(cp_cold * 1e3, 2), us_copy_same_bytes_hot=round(cp_hot * 1e3, 2),
               steps_per_s=round(n / ms_cold * 1e3),
               roofline=dict(bound="hbm", achieved=round(alg_bytes / ms_cold / 1e6, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=frac(ms_cold),
                             traffic=traffic, traffic_source=traffic_src, algorithmic_bytes=alg_bytes, hot_frac=frac(ms_hot), copy_frac=frac(cp_cold)))
    del fns, sets
    torch.cuda.empty_cache()
    # CPU side: the C port and the interpreter-bound Python form (the reference runs a Python loop) on bounded samples, 1 thread
    from oracle import gae as ogae
    rews, dones, trunc, values = host
    t = time.perf_counter()
    ogae.gae(rews, dones, trunc, values, 0.99, 0.95, 1.7, "f64")
    out["cpu_c_port_steps_per_s"] = round(n / (time.perf_counter() - t))
    m = 65536
    t = time.perf_counter()
    ogae.gae_python(rews[:m], dones[:m], trunc[:m], values[:m + 1], 0.99, 0.95, 1.7)
    out["cpu_python_loop_steps_per_s"] = round(m / (time.perf_counter() - t))
    log("gae cpu: C port over all 2,097,152 steps, Python-loop form (what the reference runs) over the first 65,536 steps, 1 thread")
    return out


def rollout_bench(learner):
    """Rollout inference at the configs[1] shape: obs [4096,107] on the host -> actions/logp on the host, per step."""
    rs = np.random.RandomState(0)
    obs = np.clip(rs.randn(N_AGENTS, OBS), -5, 5).astype(np.float32)
    q_host = torch.empty(N_AGENTS, ACT).exponential_(1)
    q_dev = q_host.cuda()
    pol = learner.policy

    def timed(fn, reps=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps

    obs_pin = torch.from_numpy(obs).pin_memory().numpy()          # the same observations in page-locked host memory
    t_parity = timed(lambda: pol.get_action(obs))                 # host-drawn Exp(1) noise (reference's CPU stream)
    t_given = timed(lambda: pol.get_action(obs, noise=q_dev))     # noise already resident
    t_pinned = timed(lambda: pol.get_action(obs_pin, noise=q_dev))
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows = pol.arena.stage_obs(torch.from_numpy(obs).cuda())
    t_kernel = time_region(lambda: pol.act_padded(rows, q_dev), 50, warm=5)   # the fused launch alone (HIP events)

    def call_us(n, reps=300):  # the reference's process-per-environment collector: 8-80 observations per get_action call
        o = obs[:n].copy()
        for _ in range(30):
            pol.get_action(o)
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            pol.get_action(o)
            ts.append(time.perf_counter() - t)
        return round(1e6 * float(np.median(ts)), 1)
    small = {"us_per_get_action_%d_obs" % n: call_us(n) for n in (8, 80)}
    log("rollout: one step = 4096 x 107 observations on the host -> actions + log-probs on the host through rlppo_discrete_step (pad + MLP + "
        "softmax + clamp + argmax(p/q) + log p in ONE launch storing into pinned host memory).  host_noise = the bit-exact parity mode (Exp(1) "
        "noise of torch's CPU generator stream drawn by librlppo's host implementation); resident_noise = noise already in HBM; pinned = the "
        "observations in page-locked memory; fused_launch = the launch alone (HIP events); us_get_action_N = wall clock of one get_action "
        "call of N host observations in the bit-exact mode (one hipGraph replay)")
    return dict(workload="4096 x 107 obs/step, 256x3 policy, 90 actions", us_get_action_8=small["us_per_get_action_8_obs"],
                us_get_action_80=small["us_per_get_action_80_obs"], obs_per_s_host_noise=round(N_AGENTS / t_parity),
                ms_per_step_host_noise=round(t_parity * 1e3, 3), obs_per_s_resident_noise=round(N_AGENTS / t_given),
                ms_per_step_resident_noise=round(t_given * 1e3, 3), ms_per_step_resident_noise_pinned_obs=round(t_pinned * 1e3, 3),
                ms_fused_launch=round(t_kernel, 4))


def cpu_baseline(seed=123, reps=2):
    """The oracle's learn() (torch-CPU eager, the reference's op sequence; kind "port") on a BOUNDED sample of the same
    workload: [r5] ONE FULL EPOCH of it -- the whole 524,288-sample configs[1] buffer, batch 524,288, 8 minibatches of 65,536, one
    optimiser step: exactly a tenth of the GPU line's 10-epoch step, sample for sample (rounds 1-4 timed a 131,072-sample slice) --
    `reps` warm repetitions per thread count, median.
    Thread counts: 16 and min(cores, 64) are both timed and both reported -- on the 256-thread GPU-box host torch-CPU with all
    hardware threads is ~20x SLOWER than with 16-64 (oversubscribed MKL/OpenMP), which would misrepresent the reference;
    `value` is the better median."""
    from oracle import nets, ppo
    cores = os.cpu_count() or 1
    n = B = N_SAMPLES
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed)
    states = torch.randn(n, OBS, generator=g).clamp_(-5, 5)
    pol0 = nets.init_mlp(OBS, HID, ACT)
    val0 = nets.init_mlp(OBS, HID, 1)
    with torch.no_grad():
        probs = nets.discrete_probs(pol0, states[:65536])
        a, lp = nets.discrete_sample(probs, torch.empty(65536, ACT).exponential_(1, generator=g))
    buf = dict(states=states, actions=a.float().repeat(n // 65536), log_probs=lp.repeat(n // 65536),
               values=torch.randn(n, generator=g), advantages=torch.randn(n, generator=g))
    per_threads, t_all = {}, time.perf_counter()
    for threads in sorted({min(cores, 16), min(cores, 64)}):
        torch.set_num_threads(threads)
        times = []
        for rep in range((reps if threads <= 16 else 1) + 1):  # the first repetition is the warm-up; the wider count (slower on this host) is timed once
            pol = [(w.clone(), b.clone()) for w, b in pol0]
            val = [(w.clone(), b.clone()) for w, b in val0]
            t = time.perf_counter()
            ppo.learn("discrete", pol, val, buf, B, MINIBATCH, 1, 0.2, 0.005, 3e-4, 3e-4, np.random.RandomState(seed))
            if rep:
                times.append(time.perf_counter() - t)
        per_threads[threads] = float(np.median(times))
    threads = min(per_threads, key=per_threads.get)
    return dict(value=round(n / per_threads[threads]), unit="samples/s", cores=threads, kind="port",
                by_threads={str(k): round(n / v) for k, v in per_threads.items()}, host_cores=cores, physical_cores=physical_cores(),
                sample="one full epoch of the GPU workload (524,288 samples, 8 minibatches, 1 optimiser step); torch-CPU oracle, median of %d reps at 16 threads (1 at 64); %.0f s"
                       % (reps, time.perf_counter() - t_all))


REF_BATCH, REF_BUFFER = 50_000, 150_000  # /root/reference: learner.py:34-53 (ppo_batch_size 50,000, minibatch = batch), example.py:74-88 (buffer 150,000)


def _ref_defaults_parity(learner, pol0, val0, opol, oval, obs, acts, logp, adv, tgt, hip_grads, report, oreport, got_p, got_v, n, B, seed):
    """The parity object of the ref_defaults leg (see the comment at its call)."""
    import contextlib
    from oracle import nets, ppo
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fp64_gate
    from rlgym_ppo_amd import _native as N
    dims_p, dims_v = [OBS] + list(HID) + [ACT], [OBS] + list(HID) + [1]

    def unflat(flat, dims):
        params, o = [], 0
        for i in range(len(dims) - 1):
            w = flat[o:o + dims[i + 1] * dims[i]].reshape(dims[i + 1], dims[i]); o += w.size
            bb = flat[o:o + dims[i + 1]]; o += bb.size
            params.append((w, bb))
        return params
    idx = np.random.RandomState(seed).permutation(n)[:B]
    n_pol = got_p.numel()
    with contextlib.redirect_stdout(sys.stderr):
        res = fp64_gate.gate(N.lib(), "discrete", pol0, val0, obs[idx], acts[idx], logp[idx], adv[idx], tgt[idx], 0.2, 0.005, 1.0,
                             (unflat(hip_grads[0][:n_pol], dims_p), unflat(hip_grads[0][n_pol:], dims_v), None), label="ref_defaults, first 50,000-row pass")
    weakest = {}
    ppo.learn64("discrete", pol0, val0, dict(states=obs, actions=acts, log_probs=logp, values=tgt, advantages=adv), B, B, 1, 0.2, 0.005, 3e-4, 3e-4,
                np.random.RandomState(seed), weakest=weakest)
    D = res["hip"]["ambiguity"] + res["cpu"]["ambiguity"] + 2e-5
    n_steps, worst, n_ill, med, mx = n // B, 0.0, 0, 0.0, 0.0
    for got, ref, weak in ((got_p, nets.flatten(opol).detach(), weakest["pol"]), (got_v, nets.flatten(oval).detach(), weakest["val"])):
        g64, r64 = got.detach().numpy().astype(np.float64), ref.numpy().astype(np.float64)
        scale = float(np.abs(r64).max())
        dd = np.abs(g64 - r64)
        allow = n_steps * 3e-4 * np.minimum(1.0, D / np.maximum(weak, 1e-300)) + 1e-5 * scale
        worst, n_ill = max(worst, float((dd / allow).max())), n_ill + int((weak < 1e-4).sum())
        med, mx = max(med, float(np.median(dd) / scale)), max(mx, float(dd.max() / scale))
    sig = lambda x: float("%.3g" % x)
    return dict(
        first_step_gradient=dict(hip_vs_fp64=sig(res["hip"]["err"]), cpu_oracle_vs_fp64=sig(res["cpu"]["err"]), relu_decisions_differing=[res["hip"]["flips"], res["cpu"]["flips"]],
                                 decisions_worth_D=sig(D), clip_edge_rows=int(len(res["edge"]))),
        params_after_3_steps=dict(worst_fraction_of_derived_allowance=sig(worst), ill_conditioned_entries=n_ill, entries=int(got_p.numel() + got_v.numel()),
                                  median_rel=sig(med), max_rel=sig(mx)),
        report_rel={k: sig(abs(report[k] - oreport[k]) / max(abs(oreport[k]), 1e-12))
                    for k in ("Policy Entropy", "Mean KL Divergence", "Value Function Loss", "SB3 Clip Fraction")},
        note="GPU vs the CPU oracle after the identical 1-epoch learn(); gradient: max|err| / max|g| per tensor against float64 under each side's own ReLU / clip-edge "
             "decisions; allowance per entry = 3 lr min(1, D / w_i) + 1e-5 max|p| (Adam's scale-free step; w_i = smallest |g_i| / max|g| over the steps, float64)")


def ref_defaults_leg(device, seed=123):
    """[r5] The reference's OWN configuration, like for like: PPOLearner.learn at ppo_batch_size = ppo_minibatch_size = 50,000
    over a 150,000-sample buffer (example.py:74-88; Learner's defaults, learner.py:34-53, have the same batch / minibatch), 256x3
    nets, obs 107, 90 actions; 1 epoch (example.py) and 10 epochs (Learner's default ppo_epochs) -- 3 and 30 optimiser steps of ONE
    50,000-row pass each (390.6 row tiles: ragged launches, no minibatch fusion, no paired launches).  The CPU oracle runs the
    IDENTICAL full workload beside it on this host's cores (same initial weights, same buffer, same numpy permutation stream), so
    the GPU/CPU pair is the first one of this repository that is not a bounded slice against a full job -- and the two results are
    compared (parameters after the 1-epoch learn(); the oracle is the checker here, never the thing shipped)."""
    from oracle import nets, ppo
    from rlgym_ppo_amd.ppo import ExperienceBuffer, PPOLearner
    import contextlib
    n, B = REF_BUFFER, REF_BATCH
    rs = np.random.RandomState(seed)
    obs = np.clip(rs.randn(n, OBS), -5, 5).astype(np.float32)
    adv, tgt = rs.randn(n).astype(np.float32), rs.randn(n).astype(np.float32)
    out = {}

    def fresh():
        torch.manual_seed(seed)
        with contextlib.redirect_stdout(sys.stderr):
            lr_ = PPOLearner(OBS, ACT, 0, HID, HID, (0.1, 1.0), B, 1, 3e-4, 3e-4, 0.2, 0.005, B, device)
        pol0 = [(l.weight.detach().cpu().clone(), l.bias.detach().cpu().clone()) for l in lr_.policy.arena.linears]
        val0 = [(l.weight.detach().cpu().clone(), l.bias.detach().cpu().clone()) for l in lr_.value_net.arena.linears]
        return lr_, pol0, val0

    learner, pol0, val0 = fresh()
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():  # actions sampled by the policy at its initial weights (ratio ~ 1 at the first step), on the host: both sides get the same
        a, lp = nets.discrete_sample(nets.discrete_probs(pol0, torch.as_tensor(obs)), torch.empty(n, ACT).exponential_(1, generator=g))
    acts, logp = a.numpy().astype(np.float32), lp.numpy()
    z = np.zeros(n, np.float32)

    def gpu_buffer():
        buf = ExperienceBuffer(n, seed, "cpu")
        buf.submit_experience(obs, acts, logp, z, obs[:1].repeat(n, 0), z, z, tgt, adv)
        return buf

    cpu_buf = dict(states=torch.as_tensor(obs), actions=torch.as_tensor(acts), log_probs=torch.as_tensor(logp), values=torch.as_tensor(tgt),
                   advantages=torch.as_tensor(adv))
    # ---- parity of the full workload (1 epoch = 3 optimiser steps), then timing of both sides
    buf = gpu_buffer()
    hip_grads = []
    learner.grad_probe = lambda gr: hip_grads.append(gr.detach().cpu().numpy().astype(np.float64)) if not hip_grads else None
    report = learner.learn(buf)
    learner.grad_probe = None
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    opol, oval = [(w.clone(), b.clone()) for w, b in pol0], [(w.clone(), b.clone()) for w, b in val0]
    t = time.perf_counter()
    oreport, _, _ = ppo.learn("discrete", opol, oval, cpu_buf, B, B, 1, 0.2, 0.005, 3e-4, 3e-4, np.random.RandomState(seed))
    cpu_1 = [time.perf_counter() - t]
    got_p = torch.nn.utils.parameters_to_vector(learner.policy.parameters()).cpu()
    got_v = torch.nn.utils.parameters_to_vector(learner.value_net.parameters()).cpu()
    # [r6] What the two float32 results may differ by, in numbers (round 5 printed a bare max |GPU - CPU| / max |CPU| = 5.7e-3 here, 500 x
    # north_star's gradient tolerance with nothing beside it).  (a) The first step's batch gradient of each side against float64 UNDER
    # ITS OWN ReLU / clip-edge decisions (tests/fp64_gate.py: what is left is arithmetic) and the worth D of those decisions.
    # (b) Adam's step lr m / (sqrt(v) + eps) is scale-free: a gradient difference d (of max|g|) moves an entry whose gradient fell to
    # w_i max|g| in some step by up to lr min(1, d / w_i) -- so every parameter is held to (steps) lr min(1, D / w_i) + 1e-5 max|p|
    # with w_i from the float64 run of the same workload (oracle/ppo.py::learn64): the worst entry's FRACTION of that allowance and
    # the number of ill-conditioned entries (w_i < 1e-4) go on the line.
    try:
        out["parity_vs_cpu_oracle"] = _ref_defaults_parity(learner, pol0, val0, opol, oval, obs, acts, logp, adv, tgt, hip_grads, report, oreport, got_p, got_v, n, B, seed)
    except Exception as e:  # noqa: BLE001 -- a failed gate goes on the line, it does not take the line with it
        out["parity_vs_cpu_oracle"] = dict(error=repr(e)[:300])
        log("ref_defaults parity failed: %r" % (e,))
    for epochs, steps in ((1, 10), (10, 3)):
        learner.n_epochs = epochs
        learner.learn(buf)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            learner.learn(buf)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / steps
        samples = epochs * (n // B) * B
        out["epochs_%d" % epochs] = dict(value=round(samples / dt, 1), unit="samples/s", ms_per_learn=round(dt * 1e3, 3),
                                         optimiser_steps=epochs * (n // B), samples_per_learn=samples,
                                         frac_of_f32_mfma_peak=round(FLOP_PER_SAMPLE * samples / dt / 1e12 / MFMA_F32_PEAK_TF, 4))
    # the CPU side: the same full workloads (1 epoch: 2 more repetitions, median of 3; 10 epochs: once)
    for _ in range(2):
        opol, oval = [(w.clone(), b.clone()) for w, b in pol0], [(w.clone(), b.clone()) for w, b in val0]
        t = time.perf_counter()
        ppo.learn("discrete", opol, oval, cpu_buf, B, B, 1, 0.2, 0.005, 3e-4, 3e-4, np.random.RandomState(seed))
        cpu_1.append(time.perf_counter() - t)
    opol, oval = [(w.clone(), b.clone()) for w, b in pol0], [(w.clone(), b.clone()) for w, b in val0]
    t = time.perf_counter()
    ppo.learn("discrete", opol, oval, cpu_buf, B, B, 2, 0.2, 0.005, 3e-4, 3e-4, np.random.RandomState(seed))
    cpu_10 = time.perf_counter() - t
    thr = torch.get_num_threads()
    out["cpu_baseline"] = dict(kind="port", cores=thr, unit="samples/s", epochs_1=round(n / float(np.median(cpu_1))), epochs_10=round(2 * n / cpu_10),
                               sample="identical workload (150,000 buffer, B = MB = 50,000): 1 epoch median of 3; multi-epoch rate over 2 epochs; torch-CPU oracle, %d threads, %.0f s"
                                      % (thr, sum(cpu_1) + cpu_10))
    out["gpu_over_cpu"] = dict(epochs_1=round(out["epochs_1"]["value"] / out["cpu_baseline"]["epochs_1"], 1),
                               epochs_10=round(out["epochs_10"]["value"] / out["cpu_baseline"]["epochs_10"], 1))
    # the launch shapes of a 50,000-row pass, isolated (the same table as kernel_breakdown, compact)
    learner._fused_rows = B
    rows, dom, _ = kernel_breakdown(learner)
    out["kernel_breakdown_50000_rows"] = [dict(kernel=r["kernel"], n=r["n"], ms=r["ms"], frac=r["frac"]) for r in rows if r["n"]]
    out["roofline"] = dict(bound="mfma", achieved=dom["tflops"], peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=round(dom["tflops"] / MFMA_F32_PEAK_TF, 4),
                           traffic=None, traffic_source="no PMC pass at this row count", kernel=dom["kernel"], ms_per_launch=dom["ms"], rows_per_launch=B)
    out["workload"] = "reference defaults: buffer 150,000, ppo_batch_size = ppo_minibatch_size = 50,000, 256x3 policy + critic, obs 107, 90 actions, fp32"
    log("ref_defaults: GPU %.2f M samples/s (1 epoch) / %.2f M (10 epochs); CPU oracle, identical workload: %.0f / %.0f samples/s; worst parameter at %.3f of its derived allowance"
        % (out["epochs_1"]["value"] / 1e6, out["epochs_10"]["value"] / 1e6, out["cpu_baseline"]["epochs_1"], out["cpu_baseline"]["epochs_10"],
           out["parity_vs_cpu_oracle"].get("params_after_3_steps", {}).get("worst_fraction_of_derived_allowance", float("nan"))))
    return out


def physical_cores():
    """Physical cores of the host (distinct (package, core) pairs in /proc/cpuinfo; os.cpu_count() counts hardware threads)."""
    try:
        pairs, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if core is not None:
                    pairs.add((phys, core))
                phys = core = None
        return len(pairs) or None
    except Exception:
        return None


def x3_leg(learner, buf, epochs, steps=3, warmup=1):
    """OPT-IN extra, never the headline: the same learn() on the same workload with rlppo_set_update_precision(2) -- fp32 data,
    losses, dW, clip, Adam and fp32-grade products, but the hidden forward / dX launches on the bf16 MFMA pipe from three bf16
    pieces per operand (csrc/gemm_split.hip; held to the same float64 gates as the fp32 precision by tests/test_gpu_kernels.py and
    tests/test_gpu_learner.py).  Also times the 256 -> 256 forward of both forms at the update's launch shape."""
    from rlgym_ppo_amd import _native as N
    from rlgym_ppo_amd.engine import set_update_precision
    L = N.lib()
    set_update_precision("x3")
    try:
        for _ in range(warmup):
            learner.learn(buf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            learner.learn(buf)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        set_update_precision("fp32")
    value = steps * epochs * BATCH / dt
    M = int(getattr(learner, "_fused_rows", MINIBATCH))
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    A = torch.randn(M, 256, device=learner._dev).clamp_(min=0)
    W = torch.randn(256, 256, device=learner._dev) * 0.05
    bias = torch.zeros(256, device=learner._dev)
    C = torch.empty(M, 256, device=learner._dev)
    bits = torch.zeros(max(int(L.rlppo_dbg_gemm_nt_bits_bytes(M, 256)), 8), dtype=torch.uint8, device=learner._dev)
    planes = torch.zeros(3 * 256 * 256, dtype=torch.bfloat16, device=learner._dev)
    N.check(L.rlppo_dbg_pack_x3(st(), P(W), 256, 256, 256, P(planes)))
    ms3 = time_region(lambda: N.check(L.rlppo_dbg_gemm_nt_x3(st(), P(A), 256, P(planes), P(bias), P(C), 256, M, 256, 256, 0, P(bits))), 20, warm_s=0.3)
    ms32 = time_region(lambda: N.check(L.rlppo_dbg_gemm_nt_bits(st(), P(A), 256, P(W), 256, P(bias), P(C), 256, M, 256, 256, 1, P(bits))), 20, warm_s=0.3)
    flop = 2 * M * 256 * 256
    log("update_x3 (opt-in): %.1f M samples/s; hidden forward %.4f ms split-bf16 (%.1f fp32-equivalent TFLOP/s) vs %.4f ms fp32 MFMA (%.1f)"
        % (value / 1e6, ms3, flop / ms3 / 1e9, ms32, flop / ms32 / 1e9))
    return dict(value=round(value, 1), unit="samples/s", ms_per_step=round(dt / steps * 1e3, 3), steps=steps,
                precision="opt-in: fp32 data and results, hidden forward / dX products as six bf16-piece MFMAs (rlppo_set_update_precision(2)); NOT the headline",
                equivalent_frac_of_f32_mfma_peak=round(FLOP_PER_SAMPLE * value / 1e12 / MFMA_F32_PEAK_TF, 4),
                fwd_hidden_ms=round(ms3, 4), fwd_hidden_ms_fp32_mfma=round(ms32, 4), fwd_hidden_tflops_fp32_equivalent=round(flop / ms3 / 1e9, 1))


BF16_MFMA_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (the 5 PF headline figure includes 2:1 sparsity)


def cfg5_rooflines(value, bf16, learner, full=True):
    """BASELINE configs[4]: 10,435,584 algorithmic flop/sample; 968 algorithmic B/sample read by the update.
    Times the three hidden-layer launch shapes of a pass (512 -> 512 at the rows the update launches) through the diagnostic
    entry points and prices each against BOTH rooflines -- the MFMA peak of its arithmetic type and 8 TB/s of HBM over its
    algorithmic bytes -- and says which one binds."""
    from rlgym_ppo_amd import _native as N
    L = N.lib()
    dev = learner._dev
    M = int(getattr(learner, "_fused_rows", MINIBATCH))
    H = 512
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    A = torch.randn(M, H, device=dev)
    A2 = torch.randn(M, H, device=dev)
    W = torch.randn(H, H, device=dev) * 0.05
    bias = torch.zeros(H, device=dev)
    C = torch.empty(M, H, device=dev)
    bits = torch.zeros(max(int(L.rlppo_dbg_gemm_nt_bits_bytes(M, H)), 8), dtype=torch.uint8, device=dev)
    dW, db = torch.zeros(H * H, device=dev), torch.zeros(H, device=dev)
    tn_ws = torch.empty(int(L.rlppo_dbg_gemm_tn_workspace_bytes(H, H, M)), dtype=torch.uint8, device=dev)
    Ab, Wb, Cb = A.bfloat16(), W.bfloat16(), torch.empty(M, H, dtype=torch.bfloat16, device=dev)
    flop = 2 * M * H * H
    dYb = torch.randn(M, H, device=dev).bfloat16()
    dXb = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
    f32_shapes = [
        ("fwd hidden 512->512 f32, +bitmask, fp32 MFMA", lambda: N.check(L.rlppo_dbg_gemm_nt_bits(st(), P(A), H, P(W), H, P(bias), P(C), H, M, H, H, 1, P(bits))),
         "f32", 4 * (2 * M * H + H * H) + M * H // 8),
        ("dX hidden 512->512 f32, bitmask, fp32 MFMA", lambda: N.check(L.rlppo_dbg_gemm_nt_bits(st(), P(A), H, P(W), H, None, P(C), H, M, H, H, 3, P(bits))),
         "f32", 4 * (2 * M * H + H * H) + M * H // 8),
        ("dW hidden 512x512 f32, + reduction, fp32 MFMA", lambda: N.check(L.rlppo_dbg_gemm_tn(st(), P(A), H, H, P(A2), H, H, P(dW), P(db), H, H, M, P(tn_ws), tn_ws.numel())),
         "f32", 4 * (2 * M * H + H * H)),
    ]
    b16_shapes = [
        ("fwd hidden 512->512 bf16: bf16 operands in memory, bf16 MFMA, fp32 accumulate; writes bf16 + ReLU bitmask",
         lambda: N.check(L.rlppo_dbg_gemm_nt_b16(st(), P(Ab), H, P(Wb), H, P(bias), None, 0, P(Cb), H, M, H, H, 1, 1, P(bits))),
         "bf16", 2 * (M * H + H * H) + 2 * M * H + M * H // 8),
        ("dX hidden 512->512 bf16: bf16 dY x bf16 W^T, rounded to bf16, masked by the bitmask",
         lambda: N.check(L.rlppo_dbg_gemm_nt_b16(st(), P(dYb), H, P(Wb), H, None, None, 0, P(dXb), H, M, H, H, 3, 2, P(bits))),
         "bf16", 2 * (M * H + H * H) + 2 * M * H + M * H // 8),
        ("dW hidden 512x512 bf16: + reduction, bf16 dY^T x bf16 X through transposing LDS reads",
         lambda: N.check(L.rlppo_dbg_gemm_tn_b16(st(), P(dYb), H, P(Ab), H, P(dW), P(db), H, H, H, H, M, P(tn_ws), tn_ws.numel())),
         "bf16", 2 * (2 * M * H) + 4 * H * H),
    ]
    # the selected precision's kernels (the first line is the roofline kernel); `full` adds the other precision's for comparison
    shapes = (b16_shapes + (f32_shapes if full else [])) if bf16 else (f32_shapes + (b16_shapes if full else []))
    rows = []
    for name, fn, kind, nbytes in shapes:
        ms = time_region(fn, 10, warm_s=0.3)
        peak = BF16_MFMA_PEAK_TF if kind == "bf16" else MFMA_F32_PEAK_TF
        f_mfma, f_hbm = flop / ms / 1e9 / peak, nbytes / ms / 1e6 / HBM_PEAK_GBS
        rows.append(dict(kernel=name.split(":")[0].split(",")[0], ms=round(ms, 4), tflops=round(flop / ms / 1e9, 1), mfma_peak=peak,
                         frac_mfma=round(f_mfma, 3), alg_mb=round(nbytes / 1e6, 1), gb_per_s=round(nbytes / ms / 1e6, 1),
                         frac_hbm=round(f_hbm, 3), binds="hbm" if f_hbm > f_mfma else "mfma"))
        log("  %-110s %8.4f ms  %7.1f TFLOP/s (%.2f of %s peak)  %7.1f GB/s (%.2f of HBM)" %
            (name[:110], ms, flop / ms / 1e9, f_mfma, kind, nbytes / ms / 1e6, f_hbm))
    fps = CFG5["flop_per_sample"]
    tf = fps * value / 1e12
    # the whole update against the MFMA peak of the precision its hidden-layer products run in (never a fraction above 1: the
    # bf16 precision is priced against 2.5 PFLOP/s, where it sits far below the MFMA roofline because its kernels are HBM-side bound)
    peak = BF16_MFMA_PEAK_TF if bf16 else MFMA_F32_PEAK_TF
    out = {"kernel_breakdown": rows,
           "update_flop_efficiency": dict(achieved=round(tf, 1), peak=peak, unit="TFLOP/s", frac=round(tf / peak, 4))}
    dom = rows[0]
    bound = dom["binds"]
    c5_traffic, c5_src = cfg5_traffic(bf16)
    out["roofline"] = dict(bound=bound, achieved=dom["gb_per_s"] if bound == "hbm" else dom["tflops"],
                           peak=HBM_PEAK_GBS if bound == "hbm" else dom["mfma_peak"], unit="GB/s" if bound == "hbm" else "TFLOP/s",
                           frac=dom["frac_hbm"] if bound == "hbm" else dom["frac_mfma"], traffic=c5_traffic, traffic_source=c5_src, kernel=dom["kernel"],
                           algorithmic_bytes=round(dom["alg_mb"] * 1e6),
                           other_bound=dict(bound="mfma" if bound == "hbm" else "hbm", frac=dom["frac_mfma"] if bound == "hbm" else dom["frac_hbm"]),
                           ms_per_launch=dom["ms"])
    log("cfg5 roofline: the hidden-layer forward of the selected update precision at the update's launch shape (M = %d rows), priced against "
        "both rooflines (algorithmic bytes: operands once, outputs once; 10,435,584 algorithmic flop/sample for the whole update); `bound` = "
        "the larger fraction.  HIP events on the launch stream, 10 launches after a 0.3 s clock ramp" % M)
    return out


def cfg5_traffic(bf16):
    """HBM bytes per launch of the configs[4] hidden-layer forward from the committed PMC passes of `bench.py --config cfg5`
    (tools/round_profile.sh: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), newest round first; None for the fp32 precision (its
    kernels are the cfg2 ones at K = 512: no pass of their own)."""
    if not bf16:
        return None, "no PMC pass (the fp32 precision runs the cfg2 kernels at K = 512)"
    mine, seen = build_id(), []
    for tag in ("r06", "r05", "r04", "r03", "r02"):
        name = tag + "_cfg5_bf16_traffic.json"
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
        if t.get("_build_id") != mine:
            seen.append("%s (build %s)" % (name, t.get("_build_id") or "unstamped"))
            continue
        for k, v in t.items():
            if "gemm_nt_b16" in k and ("<1," in k or "<1>" in k or "fwd" in k):
                return round(v["hbm_bytes"]), "build " + mine + ": profiles/" + name + PMC_NOTE
    if seen:
        return None, "stale: " + seen[0] + " was not measured on this library (build " + mine + ")"
    return None, "no committed PMC pass"


def cfg5_leg(device, steps=2, warmup=1):
    """BASELINE configs[4] on the default driver line: the same PPO-update metric on the Gaussian-policy workload (obs 231, 512x4
    nets, 524,288-sample buffer, 10 epochs per step), in the fp32 update precision (the parity mode: the reference's arithmetic)
    and in the bf16 update precision (mixed-precision training; the reference has no such mode: parity-unpinned by construction,
    checked against the repository's own restatement, tests/test_gpu_cfg5.py), `steps` timed learn() calls each, with the
    dominant kernel priced against BOTH rooflines."""
    import contextlib
    from rlgym_ppo_amd.engine import set_update_precision
    out = {}
    with contextlib.redirect_stdout(sys.stderr):
        learner, buf = build_workload(device, config="cfg5")
    for prec in ("fp32", "bf16"):
        set_update_precision(prec)
        try:
            for _ in range(warmup):
                learner.learn(buf)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                report = learner.learn(buf)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            value = steps * learner.n_epochs * BATCH / dt
            log("cfg5 %s: %.1f M samples/s" % (prec, value / 1e6))
            leg = dict(value=round(value, 1), unit="samples/s", steps=steps, ms_per_step=round(dt / steps * 1e3, 2),
                       dtype="bf16" if prec == "bf16" else "f32",
                       parity="pinned: fixtures G9 (fp32 arithmetic)" if prec == "fp32" else
                              "parity-unpinned: the reference has no mixed-precision mode",
                       kl=round(float(report["Mean KL Divergence"]), 6), value_loss=round(float(report["Value Function Loss"]), 5))
            leg.update(cfg5_rooflines(value, prec == "bf16", learner, full=False))
            out[prec] = leg
        finally:
            set_update_precision("fp32")
    out["workload"] = "configs[4]: 524,288 samples, obs 231, Gaussian 8 actions, 512x4 nets, batch 524,288, minibatch 65,536, 10 epochs/step"
    del learner, buf
    torch.cuda.empty_cache()
    return out


class BenchVectorEnv:
    """4096 agents stepping in lockstep with auto-reset (the interface VectorAgentManager drives): pre-drawn observations -- the
    leg measures the learner side of an iteration, not numpy's randn."""

    class _Space:
        def __init__(self, shape=None, n=None):
            self.shape = shape
            if n is not None:
                self.n = n

        def seed(self, s):
            pass

    def __init__(self):
        self.n_agents = N_AGENTS
        self.observation_space = BenchVectorEnv._Space(shape=(OBS,))
        self.action_space = BenchVectorEnv._Space(n=ACT)
        self._pool = (np.random.RandomState(0).randn(8, N_AGENTS, OBS) * 2 + 0.5).astype(np.float32)
        self._i = 0
        self.ep_len = 40 + (np.arange(N_AGENTS) * 7) % 100
        self.t = np.zeros(N_AGENTS, np.int64)

    def _obs(self):
        self._i += 1
        return self._pool[self._i % 8]

    def reset(self):
        self.t[:] = 0
        return self._obs()

    def step(self, actions):
        actions = np.asarray(actions, np.float32).reshape(self.n_agents, -1)
        self.t += 1
        rew = np.tanh(actions[:, 0] * 0.01).astype(np.float32)
        done = self.t >= self.ep_len
        self.t[done] = 0
        return self._obs(), rew, done.astype(np.float32), np.zeros(self.n_agents, np.float32), {"state": None}

    def close(self):
        pass


def iteration_leg(iters=3):
    """One whole iteration at the configs[1] scale through the reference's own entry points (rlgym_ppo/learner.py:257-270:
    collect_timesteps -> add_new_experience -> ppo_learner.learn): 4096 agents x 128 steps collected by the device-resident
    VectorAgentManager from a synthetic vectorised environment with the bit-exact host noise stream, the value pass + GAE +
    buffer submit on the device, and the 10-epoch update.  Median of `iters` iterations after one warm-up."""
    import contextlib
    from rlgym_ppo_amd import Learner
    with contextlib.redirect_stdout(sys.stderr):
        learner = Learner(BenchVectorEnv, vector_env=True, n_proc=1, timestep_limit=10**9, exp_buffer_size=N_SAMPLES,
                          ts_per_iteration=N_SAMPLES, ppo_epochs=10, ppo_batch_size=BATCH, ppo_minibatch_size=MINIBATCH,
                          policy_layer_sizes=HID, critic_layer_sizes=HID, checkpoints_save_folder=None,
                          checkpoint_load_folder=None, save_every_ts=10**12, log_to_wandb=False, random_seed=123)
    sync = torch.cuda.synchronize
    rows = []
    try:
        learner.ppo_learner.policy.noise_mode = "device"   # (not the reference's CPU stream: no host work per step)
        t_dev = []
        for it in range(3):
            sync(); t0 = time.perf_counter()
            learner.agent.collect_timesteps(N_SAMPLES)
            sync(); t_dev.append(time.perf_counter() - t0)
        learner.ppo_learner.policy.noise_mode = "host"
        for it in range(iters + 1):
            sync(); t0 = time.perf_counter()
            exp, _, n, _ = learner.agent.collect_timesteps(N_SAMPLES)
            sync(); t1 = time.perf_counter()
            learner.add_new_experience(exp)
            sync(); t2 = time.perf_counter()
            with contextlib.redirect_stdout(sys.stderr):
                learner.ppo_learner.learn(learner.experience_buffer)
            sync(); t3 = time.perf_counter()
            if it:
                rows.append((t1 - t0, t2 - t1, t3 - t2))
    finally:
        learner.agent.cleanup()
    c, a, l = (float(np.median([r[i] for r in rows])) * 1e3 for i in range(3))
    del learner
    torch.cuda.empty_cache()
    log("iteration: collect = 128 x (policy inference on 4096 observations with the reference's CPU noise stream + the environment's own "
        "step() + H2D of its observations); add_new_experience = value pass on 524,289 rows + GAE scan + ring-buffer submit, all on the "
        "device; learn = the headline metric's step; steps_per_s = agent-steps per second of the whole iteration; median of %d iterations" % iters)
    return dict(workload="4096 agents x 128 steps/iteration (configs[1] scale), synthetic vector env, 10-epoch update",
                collect_ms=round(c, 2), collect_ms_per_env_step=round(c / N_STEPS, 4),
                collect_ms_device_noise=round(min(t_dev[1:]) * 1e3, 2), add_new_experience_ms=round(a, 3), learn_ms=round(l, 3),
                iteration_ms=round(c + a + l, 2), steps_per_s=round(N_SAMPLES / (c + a + l) * 1e3))


def process_collect_leg(n_proc=8, timesteps=50_000, limit_s=120):
    """Collection as the reference's defaults run it (rlgym_ppo/learner.py:34-53: n_proc 8, min_inference_size 80 -> 0.9 n_proc;
    batched_agent_manager.py:180-221): 8 env WORKER PROCESSES (a 1v1 match each: two agents, 107-float observations, 90 actions;
    tools/bench_process_env.py, pre-drawn observations) speak the reference's wire format to BatchedAgentManager, which calls
    policy.get_action on whatever the ready workers handed in -- the 8-80-observation call of DESIGN 5f-5g -- until 50,000 timesteps are
    in.  One warm collection, one timed.  Bounded by an alarm: a collector whose workers died would wait on its socket for ever."""
    import contextlib
    import signal
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import bench_process_env
    from rlgym_ppo_amd import Learner

    def on_alarm(signum, frame):
        raise TimeoutError("process_collect: no result within %d s" % limit_s)

    old = signal.signal(signal.SIGALRM, on_alarm)
    signal.alarm(limit_s)
    learner = None
    try:
        with contextlib.redirect_stdout(sys.stderr):
            learner = Learner(bench_process_env.make_env, n_proc=n_proc, min_inference_size=80, timestep_limit=10**9, exp_buffer_size=150_000,
                              ts_per_iteration=timesteps, ppo_epochs=1, ppo_batch_size=50_000, ppo_minibatch_size=50_000,
                              policy_layer_sizes=HID, critic_layer_sizes=HID, checkpoints_save_folder=None, checkpoint_load_folder=None,
                              save_every_ts=10**12, log_to_wandb=False, random_seed=123)
        pol = learner.ppo_learner.policy
        calls = []
        inner = pol.get_action

        check = dict(n=0, bad=0, s=0.0)

        def counted(obs, *a, **k):
            # [r6] every 100th call is made twice from the same generator state: as the collector makes it (host window, late noise,
            # completion words) and through the general path (explicit copies, stream synchronisation) -- same actions, same
            # log-probabilities, same generator state afterwards, or it is counted as a mismatch
            spot = len(calls) % 100 == 37 and not a and not k
            if spot:
                st0 = torch.get_rng_state()
            t = time.perf_counter()
            out = inner(obs, *a, **k)
            t1 = time.perf_counter()
            calls.append((len(obs), t1 - t))
            if spot:
                after = torch.get_rng_state()
                torch.set_rng_state(st0)
                pol.act_graphs = False
                try:
                    ref = inner(obs)
                finally:
                    pol.act_graphs = True
                ok = torch.equal(torch.as_tensor(out[0]), torch.as_tensor(ref[0])) and torch.equal(torch.as_tensor(out[1]), torch.as_tensor(ref[1])) \
                    and torch.equal(after, torch.get_rng_state())
                check["n"] += 1
                check["bad"] += 0 if ok else 1
                check["s"] += time.perf_counter() - t1
            return out

        learner.agent.collect_timesteps(4_000)
        pol.get_action = counted
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        exp, _, n, _ = learner.agent.collect_timesteps(timesteps)
        dt = time.perf_counter() - t0
        # ... and the rest of the reference's iteration on what was collected (learner.py:157-162: add_new_experience = value pass + GAE +
        # buffer submit, then PPOLearner.learn at its defaults' batch = minibatch = 50,000, 1 epoch as in example.py:74-88)
        # -- in its steady state: the 150,000-sample buffer full (three collections), one learn() behind it to warm the launches
        pol.get_action = inner
        for _ in range(3):
            learner.add_new_experience(exp)
        learner.ppo_learner.learn(learner.experience_buffer)
        torch.cuda.synchronize()
        t_add = time.perf_counter()
        learner.add_new_experience(exp)
        torch.cuda.synchronize()
        t_learn = time.perf_counter()
        lrep = learner.ppo_learner.learn(learner.experience_buffer)
        torch.cuda.synchronize()
        t_end = time.perf_counter()
        add_ms, learn_ms = (t_learn - t_add) * 1e3, (t_end - t_learn) * 1e3
        rows = np.array([c[0] for c in calls])
        secs = np.array([c[1] for c in calls])
        log("process_collect: %d worker processes, %d timesteps in %.3f s; %d get_action calls of %.1f observations on average, %.1f us each "
            "(median), %.1f %% of the collection's wall clock inside get_action" % (n_proc, n, dt, len(calls), rows.mean(), 1e6 * np.median(secs),
                                                                                   100 * secs.sum() / dt))
        return dict(workload="reference defaults: %d env worker processes x 2 agents (obs 107, 90 actions, 256x3 policy) through BatchedAgentManager "
                             "and its wire format, %d timesteps" % (n_proc, timesteps),
                    n_proc=n_proc, timesteps=int(n), seconds=round(dt, 3), steps_per_s=round(n / dt), get_action_calls=len(calls),
                    mean_obs_per_call=round(float(rows.mean()), 1), us_per_get_action_median=round(1e6 * float(np.median(secs)), 1),
                    frac_of_wall_in_get_action=round(float(secs.sum() / dt), 3),
                    collector="C++ (rlppo_collector_*)" if getattr(learner.agent, "_native", None) is not None else "Python",
                    # the whole iteration of the reference's loop on this collection: collect + add_new_experience + learn (1 epoch)
                    add_new_experience_ms=round(add_ms, 2), learn_ms=round(learn_ms, 2), iteration_steps_per_s=round(n / (dt + (add_ms + learn_ms) / 1e3)),
                    # the small call's transport counters over this collection (ppo/_mlp.py::ActGraph) and the 1 % spot check
                    transport={k: int(sum(getattr(g, k) for g in pol._graphs.values())) for k in ("calls", "polled", "poll_timeouts", "late_retries", "stale_relaunches")},
                    spot_checked=check["n"], spot_check_mismatches=check["bad"], spot_check_s=round(check["s"], 4))
    except Exception as e:  # noqa: BLE001 -- a leg that fails must not take the line with it
        log("process_collect failed: %r" % (e,))
        return dict(error=repr(e))
    finally:
        signal.alarm(0)
        signal.signal(signal.SIGALRM, old)
        if learner is not None:
            with contextlib.suppress(Exception):
                learner.agent.cleanup()


LINE_LIMIT = 11000  # bytes of stdout the driver's record keeps whole (round 5's 15.5 KB line lost its head)


def fit_line(out, limit=LINE_LIMIT):
    """The JSON line, compact, within `limit` bytes: prose that only explains (notes, method descriptions -- they are in DESIGN.md and on
    stderr) goes first, then long strings are cut, then whole detail tables; numbers are never dropped before prose is."""
    dumps = lambda o: json.dumps(o, separators=(",", ":"))
    line = dumps(out)
    if len(line) <= limit:
        return line
    out = json.loads(line)
    limit -= 160  # (room for the `line_fitted` remark below)

    def walk(o, fn):
        if isinstance(o, dict):
            for k in list(o):
                if fn(o, k):
                    continue
                walk(o[k], fn)
        elif isinstance(o, list):
            for v in o:
                walk(v, fn)

    def drop_keys(names):
        def fn(d, k):
            if k in names:
                del d[k]
                return True
            return False
        return fn

    def cut_strings(n):
        def fn(d, k):
            if isinstance(d[k], str) and len(d[k]) > n:
                d[k] = d[k][:n - 3] + "..."
            return False
        return fn
    steps = [lambda: walk(out, drop_keys({"note", "method", "how", "sample_note"})),
             lambda: walk(out, drop_keys({"exec_tflops", "gflop"})),   # (derivable: flop per launch is the shape, executed = padded shape)
             lambda: walk(out, cut_strings(160)), lambda: walk(out, cut_strings(90)),
             lambda: walk(out, drop_keys({"source", "traffic_source"})),
             lambda: out.get("process_collect", {}).pop("n_proc_32", None),
             lambda: walk(out, cut_strings(48)),
             lambda: out.pop("update_x3", None), lambda: out.pop("non_gemm_tail", None), lambda: out.pop("iteration", None)]
    for step in steps:
        step()
        line = dumps(out)
        if len(line) <= limit:
            break
    out["line_fitted"] = "shortened to fit %d bytes of the driver's record; the full objects are on stderr / in profiles/" % (limit + 160)
    return dumps(out)


# ---------------------------------------------------------------------------------------------------- main
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--no-extras", action="store_true", help="skip roofline / gae / rollout / cpu_baseline legs")
    ap.add_argument("--config", choices=("cfg2", "cfg5"), default="cfg2",
                    help="cfg2 = BASELINE configs[1] (the headline metric); cfg5 = configs[4] shape (Gaussian policy, obs 231, "
                         "512x4 nets) -- update throughput only, no extra legs")
    ap.add_argument("--precision", choices=("fp32", "bf16", "x3"), default="fp32",
                    help="update precision: fp32 (parity mode, default), bf16 = bf16-operand forward / fp32 master, accumulate "
                         "and backward (BASELINE configs[4]; rlppo_set_update_precision), x3 = fp32 data and results with the hidden "
                         "forward / dX products on the bf16 MFMA pipe from three-piece operands (opt-in, DESIGN 4.5)")
    ap.add_argument("--allreduce", choices=("torch", "direct", "ab"), default="torch",
                    help="N > 1: gradient exchange through torch.distributed (RCCL; the default and the product's default), through "
                         "librlppo's own RCCL communicator, or 'ab' = the timed region uses torch.distributed, the JSON line is printed, "
                         "and a short A/B of both routes follows on stderr (opt-in: the direct communicator has never run with more "
                         "than one rank on this pool, so it stays out of the line the driver records)")
    return ap.parse_args(argv)


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: THIS process starts the N ranks itself, as fresh
    children of torch.distributed.run, and relays rank 0's JSON line.  It makes no GPU call of its own (a process that has
    initialised HIP must not be the parent of the ranks' exec chain on this pool), so everything below the import of torch is
    host-only here."""
    import signal
    import subprocess
    # --standalone: the launcher picks its own free rendezvous port on 127.0.0.1 (no probe-then-bind window)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(args.gpus, 1))))
    log(f"bench.py: starting {args.gpus} ranks: {' '.join(cmd[1:8])} ...  (parent initialised HIP: {torch.cuda.is_initialized()})")
    limit = float(os.environ.get("RLPPO_BENCH_TIMEOUT", 1500))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, start_new_session=True)
    try:
        stdout, _ = proc.communicate(timeout=limit)
    except subprocess.TimeoutExpired:  # a rank that hangs must not hang the parent: end the whole job, report, fail
        log(f"bench.py: the {args.gpus}-rank job did not finish within {limit:.0f} s; killing its process group")
        os.killpg(proc.pid, signal.SIGKILL)
        stdout, _ = proc.communicate()
        print(stdout, file=sys.stderr)
        return 124
    line = None
    for out in stdout.splitlines():
        if out.startswith("{") and '"metric"' in out:
            line = out
        else:
            log(out)
    rc = proc.returncode
    if line is not None:
        print(line, flush=True)
    return rc if rc != 0 else (0 if line is not None else 1)


def allreduce_ab(learner, buf, steps, world, device, dist):
    """N > 1 only, after the timed region: the same learn() steps with the gradient exchange through librlppo's own RCCL
    communicator (rlppo_allreduce on the compute stream) next to torch.distributed's, plus the exchange alone in microseconds
    per optimiser step (the 1.37 MB [grad_policy | grad_value] arena, 200 back-to-back collectives)."""
    from rlgym_ppo_amd import dp
    out = {}
    grad = learner._grad_all
    scratch = torch.zeros_like(grad)
    for name in ("torch", "direct"):
        dp.set_allreduce_backend(name)
        try:
            for _ in range(10):
                dp.all_reduce_sum(scratch, dist)
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                dp.all_reduce_sum(scratch, dist)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / 200 * 1e6
            learner.learn(buf)
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                learner.learn(buf)
            dist.barrier()
            torch.cuda.synchronize()
            t = torch.tensor([time.perf_counter() - t0, us], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            out[name] = dict(us_per_allreduce=round(float(t[1]), 1), ms_per_step=round(float(t[0]) / steps * 1e3, 3),
                             samples_per_s=round(steps * learner.n_epochs * BATCH / float(t[0]), 1))
        except Exception as e:  # noqa: BLE001 -- the A/B must never take the headline number down with it
            out[name] = dict(error=str(e)[:200])
    dp.set_allreduce_backend("torch")
    out["bytes"] = grad.numel() * 4
    out["note"] = "same ranks, same workload, after the timed region; `value` above is the torch.distributed run"
    return out


def main():
    args = parse_args()
    torchrun = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus > 1 and not torchrun:
        sys.exit(spawn_ranks(args))

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE")
    # RLPPO_BENCH_DRYRUN=1: exercise the N > 1 code path on a single-GPU box (all ranks on cuda:0, gloo); numbers from
    # such a run are meaningless and are tagged as such.  =2: rendezvous + one CPU all-reduce only (no GPU at all: the
    # launch-path check of tests/test_bench_spawn.py).  The driver's real runs use one rank per GPU over RCCL.
    dry = os.environ.get("RLPPO_BENCH_DRYRUN", "")
    dryrun = dry in ("1", "2")
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this pool (exported there anyway)
    if dry == "2":
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ones = torch.ones(1)
        dist.all_reduce(ones)
        if rank == 0:
            print(json.dumps({"metric": "ppo_update_samples_per_sec", "value": None, "unit": "samples/s", "n_gpus": int(ones.item()),
                              "steps": args.steps, "warmup": args.warmup, "data": "none (DRY RUN 2: launch path only, no GPU work)",
                              "hip_initialised": torch.cuda.is_initialized()}), flush=True)
        dist.destroy_process_group()
        return
    if dryrun:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if world > 1:
        import faulthandler
        import threading
        faulthandler.enable()  # a fault in a rank leaves a trace and kills it (the launcher then tears the job down): no handler games
        # Watchdog around the rendezvous + communicator set-up + the first collective: if they do not finish within
        # RLPPO_INIT_TIMEOUT seconds (default 90: RCCL's 8-rank bootstrap takes 5-20 s on a cold node) this rank says why in one
        # line and exits non-zero instead of hanging until the driver's own limit.  (A thread + os._exit in THIS fresh process: a
        # process that has touched the GPU is never re-exec'd.)
        init_done = threading.Event()
        limit = float(os.environ.get("RLPPO_INIT_TIMEOUT", 90))

        def init_watchdog():
            if not init_done.wait(limit):
                log(f"bench.py rank {rank}/{world}: init_process_group / first all-reduce did not finish within {limit:.0f} s "
                    f"(backend {'gloo' if dryrun else 'nccl'}, MASTER_ADDR={os.environ.get('MASTER_ADDR')}, MASTER_PORT={os.environ.get('MASTER_PORT')}, "
                    f"device {device}, HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}); exiting 4")
                os._exit(4)
        threading.Thread(target=init_watchdog, daemon=True).start()
        t_init = time.perf_counter()
        if dryrun:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
        t_pg = time.perf_counter() - t_init
        from rlgym_ppo_amd import dp
        dp.set_allreduce_backend("direct" if args.allreduce == "direct" else "torch")
        # the number of ranks the collective really spans: a sum of ones through the backend the run uses (the first collective
        # of the job: RCCL builds its rings / trees here)
        ones = torch.ones(1, device=device)
        t_first = time.perf_counter()
        dist.all_reduce(ones)
        n_seen = int(ones.item())
        t_first = time.perf_counter() - t_first
        # the exchange the update performs (the 1.37 MB [grad_policy | grad_value] arena), warm, before anything is timed
        probe = torch.zeros(341_851, device=device)
        for _ in range(3):
            dp.all_reduce_sum(probe, dist)
        torch.cuda.synchronize()
        t_ar = time.perf_counter()
        for _ in range(20):
            dp.all_reduce_sum(probe, dist)
        torch.cuda.synchronize()
        t_ar = (time.perf_counter() - t_ar) / 20
        init_done.set()
        log(f"bench.py rank {rank}/{world} on {device}: init_process_group {t_pg:.2f} s, first all-reduce {t_first * 1e3:.1f} ms, n_seen {n_seen}, "
            f"1.37 MB gradient all-reduce {t_ar * 1e6:.0f} us warm (backend {'gloo' if dryrun else 'nccl=RCCL'}, route {args.allreduce if args.allreduce != 'ab' else 'torch'})")
        allreduce_us = round(t_ar * 1e6, 1)
    else:
        n_seen = 1
        allreduce_us = None

    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        learner, buf = build_workload(device, config=args.config)
    learner.n_epochs = args.epochs
    if args.precision != "fp32":
        from rlgym_ppo_amd.engine import set_update_precision
        set_update_precision(args.precision)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        learner.learn(buf)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        report = learner.learn(buf)
    barrier()
    dt = time.perf_counter() - t0
    per_rank = None
    if world > 1:
        # every rank's own time travels to every rank (a sum of one-hot vectors): the line reports the MAX (the contract) and the
        # spread, so a straggler is visible in the driver's record
        t = torch.zeros(world, dtype=torch.float64, device=device)
        t[rank] = dt
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        ts = [float(x) for x in t.tolist()]
        dt = max(ts)
        per_rank = dict(min=round(min(ts) / args.steps * 1e3, 3), max=round(dt / args.steps * 1e3, 3),
                        all=[round(x / args.steps * 1e3, 3) for x in ts])
        if rank == 0:
            log("per-rank ms per step: " + ", ".join("%.3f" % x for x in per_rank["all"]) + " (value uses the max)")

    samples = args.steps * args.epochs * (N_SAMPLES // BATCH) * BATCH
    value = samples / dt
    bf16 = args.precision == "bf16"
    if args.config == "cfg2":
        workload = "configs[1]: 4096 agents x 128 steps = 524,288-sample buffer, obs 107 f32, 90 discrete actions, 256x3 policy + critic; batch 524,288, minibatch 65,536"
    else:
        workload = ("configs[4]: 524,288-sample buffer, obs 231 f32, Gaussian 8 actions, 512x4 policy + critic, " +
                    ("bf16 update precision (bf16 MFMA operands forward and backward, fp32 accumulate / loss / dW / Adam / master weights)" if bf16 else "fp32 update") +
                    "; batch 524,288, minibatch 65,536")
    slices_per_rank = 8 // world if 8 % world == 0 else 1
    out = {
        "metric": "ppo_update_samples_per_sec", "value": round(value, 1), "unit": "samples/s", "n_gpus": n_seen,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "bf16" if bf16 else ("f32 (hidden forward / dX products split into bf16 pieces, opt-in)" if args.precision == "x3" else "f32"),
        "data": "synthetic" + (" (DRY RUN: all ranks on one GPU over gloo, not a measurement)" if dryrun else ""),
        "config": {"workload": workload, "epochs_per_step": args.epochs, "samples_per_step": args.epochs * BATCH,
                   "parallelism": f"dp{world}: {slices_per_rank} minibatch slice(s) per rank and pass, 1 RCCL all-reduce per optimiser step"
                                  + (f" ({args.allreduce if args.allreduce != 'ab' else 'torch'}.distributed)" if world > 1 else ""),
                   "last_report": {k: (round(v, 5) if isinstance(v, float) else v) for k, v in report.items()}},
    }
    if world > 1 and args.config == "cfg2":  # achieved fraction of the MFMA roofline of the whole job (N x peak)
        out["update_flop_efficiency"] = dict(
            achieved=round(FLOP_PER_SAMPLE * value / 1e12, 2), peak=round(MFMA_F32_PEAK_TF * world, 1), unit="TFLOP/s",
            frac=round(FLOP_PER_SAMPLE * value / 1e12 / (MFMA_F32_PEAK_TF * world), 4))  # 1,931,776 algorithmic flop/sample x whole-job samples/s vs N x the fp32 MFMA peak
    if args.config == "cfg5":
        pass  # kernel lines are added below (rank 0, one GPU)

    emitted = []

    def emit():
        if rank == 0 and not emitted:
            emitted.append(1)
            line = fit_line(out)
            if "line_fitted" in line:
                log("bench.py: the full line before it was shortened:\n" + json.dumps(out))
            print(line, flush=True)

    if world > 1:
        out["allreduce_us"] = allreduce_us  # the update's 1.37 MB exchange alone, warm, measured before the timed region (rank 0's view)
        out["per_rank_ms_per_step"] = per_rank
    if world > 1 and args.allreduce == "ab" and dry != "2":  # opt-in (the one-GPU dry run walks through it too: both legs then go through gloo)
        # The direct communicator has its first multi-rank run here, so the JSON line is printed FIRST and the A/B reports on
        # stderr.  A watchdog thread ends the process if the A/B has not finished within 120 s or as soon as SIGTERM arrives (the
        # launcher tearing the job down because another rank died) -- SIGTERM only: a fault keeps its default disposition (the rank
        # dies with faulthandler's trace, the launcher ends the job).
        import select, signal, socket, threading
        emit()
        rd, wr = socket.socketpair()
        wr.setblocking(False)
        signal.signal(signal.SIGTERM, lambda *_: None)
        signal.set_wakeup_fd(wr.fileno(), warn_on_full_buffer=False)

        def bail():
            ready, _, _ = select.select([rd], [], [], 120.0)
            got = rd.recv(16) if ready else b""
            if got[:1] == b"\0":  # the A/B finished
                return
            log("exchange A/B abandoned: " + ("terminated by signal %d" % got[0] if got else "not finished within 120 s"))
            os._exit(3)  # the headline line is out, but this process is stuck in (or was torn out of) a collective: not a clean exit
        threading.Thread(target=bail, daemon=True).start()
        ab = allreduce_ab(learner, buf, max(1, min(args.steps, 5)), world, device, dist)
        wr.send(b"\0")
        signal.set_wakeup_fd(-1)
        signal.signal(signal.SIGTERM, signal.SIG_DFL)
        if rank == 0:
            log("exchange A/B: " + json.dumps(ab))
    if rank == 0 and world == 1 and not args.no_extras and args.config == "cfg5":
        out.update(cfg5_rooflines(value, bf16, learner))
    if rank == 0 and world == 1 and not args.no_extras and args.config == "cfg2":
        rows, dom, M = kernel_breakdown(learner)
        for r in rows:
            log("  %-24s x%d  %8.4f ms  %7.2f TFLOP/s (algorithmic), %7.2f executed" % (r["kernel"], r["n"], r["ms"], r["tflops"], r["exec_tflops"]))
        traffic, traffic_src = pmc_traffic_for(dom["kernel"])
        log("roofline: dominant kernel of the timed region by total time; achieved = algorithmic flop per launch / mean launch duration (HIP "
            "events on the launch stream, 20 launches after a 0.3 s clock ramp, same shape as in the update: M = %d rows per pass); traffic = "
            "HBM bytes per launch from rocprofv3 PMC (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE): %s.  update_flop_efficiency = 1,931,776 "
            "algorithmic flop/sample x measured samples/s (whole learn(), host shuffle included)" % (M, traffic_src))
        out["roofline"] = dict(bound="mfma", achieved=dom["tflops"], peak=MFMA_F32_PEAK_TF, unit="TFLOP/s",
                               frac=round(dom["tflops"] / MFMA_F32_PEAK_TF, 4), traffic=traffic, traffic_source=traffic_src, kernel=dom["kernel"],
                               algorithmic_gflop_per_launch=dom["gflop"], ms_per_launch=dom["ms"], rows_per_launch=M)
        out["update_flop_efficiency"] = dict(achieved=round(FLOP_PER_SAMPLE * value / 1e12, 2), peak=MFMA_F32_PEAK_TF, unit="TFLOP/s",
                                             frac=round(FLOP_PER_SAMPLE * value / 1e12 / MFMA_F32_PEAK_TF, 4))
        out["kernel_breakdown"] = rows

        def leg(name, fn, *a, **k):
            """One extra leg: whatever it raises goes ON the line (and to stderr), it never takes the line with it -- the headline,
            its roofline and the CPU baseline must reach the driver even when a side leg breaks."""
            try:
                out[name] = fn(*a, **k)
            except Exception as e:  # noqa: BLE001
                import traceback
                log("bench.py: leg %s failed:\n%s" % (name, traceback.format_exc()))
                out[name] = dict(error=repr(e)[:300])

        leg("non_gemm_tail", non_gemm_tail)
        leg("update_x3", x3_leg, learner, buf, args.epochs)
        leg("gae", gae_bench)
        leg("rollout", rollout_bench, learner)
        del learner, buf  # the other legs build their own workloads
        torch.cuda.empty_cache()
        leg("iteration", iteration_leg)
        leg("cfg5", cfg5_leg, device)
        leg("cpu_baseline", cpu_baseline)
        leg("ref_defaults", ref_defaults_leg, device)
        # (last: 8 + 32 worker processes come and go here -- nothing that is timed runs beside their start-up or their exit)
        leg("process_collect", process_collect_leg, 8)                   # learner.py:34-53's default
        try:
            out["process_collect"]["n_proc_32"] = process_collect_leg(32)    # example.py:74-88
        except Exception as e:  # noqa: BLE001
            out["process_collect"]["n_proc_32"] = dict(error=repr(e)[:300])
        # The scalars a reader wants first, once more at the END of the line (a truncated record keeps its tail)

        def at(*path):
            cur = out
            for k in path:
                if not isinstance(cur, dict) or k not in cur:
                    return None
                cur = cur[k]
            return cur
        g_traffic, g_alg = at("gae", "roofline", "traffic"), at("gae", "roofline", "algorithmic_bytes")
        out["summary"] = dict(
            ppo_update_samples_per_s=out["value"], update_frac_of_f32_mfma_peak=out["update_flop_efficiency"]["frac"],
            dominant_kernel_frac=out["roofline"]["frac"],
            gae_us=at("gae", "us_per_scan"), gae_frac_hbm_cold=at("gae", "roofline", "frac"), gae_frac_hbm_hot=at("gae", "roofline", "hot_frac"),
            gae_traffic_ratio=round(g_traffic / g_alg, 3) if g_traffic and g_alg else None,
            rollout_ms_host_noise=at("rollout", "ms_per_step_host_noise"), rollout_ms_resident=at("rollout", "ms_per_step_resident_noise"),
            us_get_action_8=at("rollout", "us_get_action_8"), us_get_action_80=at("rollout", "us_get_action_80"),
            collect_ms=at("iteration", "collect_ms"), iteration_steps_per_s=at("iteration", "steps_per_s"),
            process_collect_steps_per_s=at("process_collect", "steps_per_s"),
            cfg5_fp32_samples_per_s=at("cfg5", "fp32", "value"), cfg5_bf16_samples_per_s=at("cfg5", "bf16", "value"),
            cfg5_bf16_update_frac_of_bf16_peak=at("cfg5", "bf16", "update_flop_efficiency", "frac"),
            cpu_port_samples_per_s=at("cpu_baseline", "value"), update_x3_optin_samples_per_s=at("update_x3", "value"),
            ref_defaults_samples_per_s_1_epoch=at("ref_defaults", "epochs_1", "value"),
            ref_defaults_samples_per_s_10_epochs=at("ref_defaults", "epochs_10", "value"),
            ref_defaults_cpu_samples_per_s_10_epochs=at("ref_defaults", "cpu_baseline", "epochs_10"),
            ref_defaults_worst_fraction_of_allowance=at("ref_defaults", "parity_vs_cpu_oracle", "params_after_3_steps", "worst_fraction_of_derived_allowance"),
            legs_failed=[k for k, v in out.items() if isinstance(v, dict) and "error" in v])
    emit()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
