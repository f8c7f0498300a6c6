"""CPU checks of the boundary: the C-ABI library loads and exports every symbol include/rlppo.h declares (no compute
calls -- there is no GPU here), the ctypes table covers the header, the host-only entry points work, and the oracle
is imported only where the rules allow."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "rlppo.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rlppo_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    from rlgym_ppo_amd import _native as N
    L = N.lib()
    names = header_functions()
    assert len(names) >= 20
    raw = ctypes.CDLL(N.LIB_PATH)
    for name in names:
        assert hasattr(raw, name), f"{name} declared in include/rlppo.h but not exported by librlppo.so"
        assert name in N.SIGNATURES, f"{name} has no ctypes signature in rlgym_ppo_amd/_native.py"
    assert sorted(N.SIGNATURES) == names, "ctypes table and header disagree"
    assert L.rlppo_abi_version() == N.ABI_VERSION


def test_layout_queries_host_only():
    from rlgym_ppo_amd import _native as N
    L = N.lib()
    # first-layer contraction: padded to 16 where that saves >= 10 % over 64 (107 -> 112), else to 64 (231 -> 256: bf16 kernels)
    assert [L.rlppo_padded_width(d) for d in (1, 13, 32, 33, 64, 100, 107, 112, 113, 231, 256)] == [16, 16, 32, 48, 64, 112, 112, 112, 128, 256, 256]
    assert [L.rlppo_padded_out(d) for d in (1, 16, 21, 33, 90, 96, 97, 256, 300)] == [32, 32, 32, 64, 96, 96, 128, 256, 384]
    d = N.dims_array([107, 256, 256, 256, 90])
    assert L.rlppo_flat_floats(d, 4) == 182362                       # cfg2 policy parameter count
    assert L.rlppo_flat_floats(N.dims_array([107, 256, 256, 256, 1]), 4) == 159489
    assert L.rlppo_flat_floats(N.dims_array([231, 512, 512, 512, 512, 16]), 5) == 914960  # cfg5 policy
    assert L.rlppo_packed_floats(d, 4) == 2 * (256 * 112 + 256 * 256 * 2 + 96 * 256) + 256 * 3 + 96
    assert L.rlppo_flat_floats(N.dims_array([0, 4]), 1) == -1 and b"dims" in L.rlppo_last_error()
    # argument errors come back as codes + message, never as exceptions across the ABI
    assert L.rlppo_net_pack(None, N.dims_array([4, 4]), 99, None, None) == 1001
    assert b"n_layers" in L.rlppo_last_error()
    with pytest.raises(RuntimeError, match="librlppo error 1001"):
        N.check(L.rlppo_net_pack(None, N.dims_array([4, 4]), 99, None, None))


def test_minibatch_args_struct_matches_header_field_order():
    from rlgym_ppo_amd import _native as N
    src = open(os.path.join(ROOT, "include", "rlppo.h")).read()
    body = src[src.index("typedef struct rlppo_minibatch_args {"):src.index("} rlppo_minibatch_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";")[:-1]:
        decl = decl.split("{")[-1].strip()
        if not decl:
            continue
        for part in decl.split(","):
            fields.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", part)[-1])
    assert fields == [f[0] for f in N.MinibatchArgs._fields_]


def test_mt19937_permutation_is_numpys_legacy_stream(golden):
    from rlgym_ppo_amd import _native as N
    from rlgym_ppo_amd.engine import LegacyPermutation
    g = golden("g6_shuffle")
    rng = np.random.RandomState(123)
    perm = LegacyPermutation(rng)
    p1 = perm.permutation(524288)
    assert np.array_equal(p1[:64], g["perm524288_head"])
    assert np.bitwise_xor.reduce(p1 * np.arange(1, p1.size + 1)) == g["perm524288_xor"]
    assert np.sum((p1 * (np.arange(p1.size) % 1000003)) % 2147483647) == g["perm524288_wsum"]
    assert np.array_equal(perm.permutation(524288)[:64], g["perm524288_second_head"])
    # generator object stays in sync with numpy's own implementation, for every size class incl. 0/1/2 and non-2^k
    ref = np.random.RandomState(7)
    mine = LegacyPermutation(np.random.RandomState(7))
    for n in (0, 1, 2, 3, 5, 64, 1000, 4097, 70000):
        assert np.array_equal(mine.permutation(n), ref.permutation(n)), n
    assert mine.rng.randint(1 << 30) == ref.randint(1 << 30)
    # seeding entry point == RandomState(seed)
    st = np.empty(625, np.uint32)
    N.lib().rlppo_mt19937_seed(st.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), 123)
    assert np.array_equal(st[:624], np.random.RandomState(123).get_state()[1]) and st[624] == 624


def test_two_phase_permutation_and_speculative_pipeline():
    """rlppo_mt19937_draw_targets + rlppo_apply_swap_targets == RandomState.permutation (stream and result), and the
    look-ahead pipeline never changes the observable stream: requests of another size, foreign draws from the generator
    and re-seeding between requests all fall back to the generator's current state."""
    from rlgym_ppo_amd import _native as N
    from rlgym_ppo_amd.engine import LegacyPermutation
    L = N.lib()
    ref = np.random.RandomState(11)
    st = np.empty(625, np.uint32)
    st[:624], st[624] = ref.get_state()[1], ref.get_state()[2]
    for n in (0, 1, 2, 3, 8, 9, 10, 17, 255, 256, 257, 1000, 4097, 70000, 524288):
        targets = np.full(max(n - 1, 0) + 8, 0xDEADBEEF, np.uint32)
        out = np.empty(n, np.int64)
        N.check(L.rlppo_mt19937_draw_targets(st.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), n, ctypes.c_void_p(targets.ctypes.data)))
        N.check(L.rlppo_apply_swap_targets(n, ctypes.c_void_p(targets.ctypes.data), ctypes.c_void_p(out.ctypes.data)))
        assert np.array_equal(out, ref.permutation(n)), n
        assert (targets[max(n - 1, 0):] == 0xDEADBEEF).all()                       # nothing written past the n-1 targets
        assert (targets[:max(n - 1, 0)] <= np.arange(n - 1, 0, -1)).all()          # j_i <= i
        assert np.array_equal(st[:624], ref.get_state()[1]) and st[624] == ref.get_state()[2], n
    bad = np.array([5, 0, 0], np.uint32)                                            # j > i is rejected, not applied
    assert L.rlppo_apply_swap_targets(4, ctypes.c_void_p(bad.ctypes.data), ctypes.c_void_p(np.empty(4, np.int64).ctypes.data)) != 0

    for lookahead in (0, 1, 3):
        ref = np.random.RandomState(5)
        rng = np.random.RandomState(5)
        mine = LegacyPermutation(rng, lookahead=lookahead)
        for n in (1000, 1000, 1000, 1000, 1000, 1000, 777, 777, 1000):              # size changes drop the speculation
            assert np.array_equal(mine.permutation(n), ref.permutation(n)), (lookahead, n)
        assert rng.randint(1 << 30) == ref.randint(1 << 30)                         # a foreign draw ...
        assert np.array_equal(mine.permutation(1000), ref.permutation(1000))        # ... is seen by the next request
        assert np.array_equal(mine.permutation(1000), ref.permutation(1000))
        rng.seed(99), ref.seed(99)                                                  # re-seeding as well
        for _ in range(4):
            assert np.array_equal(mine.permutation(4096), ref.permutation(4096))
        assert np.array_equal(rng.get_state()[1], ref.get_state()[1]) and rng.get_state()[2] == ref.get_state()[2]
        assert rng.standard_normal() == ref.standard_normal()
        mine.close()


def test_oracle_is_only_imported_where_allowed():
    allowed = {"bench.py", "__graft_entry__.py"}
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    offenders = []
    for dirpath, _, files in os.walk(ROOT):
        rel = os.path.relpath(dirpath, ROOT)
        if rel.startswith((".git", "gpurun_out", "scratch", "tests", "oracle")):
            continue
        for f in files:
            if f.endswith(".py"):
                path = os.path.join(dirpath, f)
                if pat.search(open(path).read()) and os.path.relpath(path, ROOT) not in allowed:
                    offenders.append(os.path.relpath(path, ROOT))
    assert offenders == [], offenders
    # the product package does not even mention it
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rlgym_ppo_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp")):
                assert "oracle" not in open(os.path.join(dirpath, f)).read().replace("oracle/gae_oracle.c mode 0", ""), f


def test_product_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rlgym_ppo_amd.ppo import PPOLearner
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        PPOLearner(107, 90, 0, (32,), (32,), (0.1, 1.0), 64, 1, 3e-4, 3e-4, 0.2, 0.005, 64, "cpu")
    with pytest.raises(RuntimeError):
        PPOLearner(107, 90, 0, (32,), (32,), (0.1, 1.0), 64, 1, 3e-4, 3e-4, 0.2, 0.005, 64, "cuda:0")


def test_welford_matches_reference(golden):
    from rlgym_ppo_amd.util import WelfordRunningStat
    g = golden("g7_welford")
    st = WelfordRunningStat(1)
    rets = g["returns"]
    assert np.array_equal(st.std, g["std_initial"]) and np.array_equal(st.mean, g["mean_initial"])
    st.increment(list(rets[:1]), 1)
    assert np.array_equal(st.std, g["std_after1"])
    st.increment(list(rets[1:150]), 149)
    assert np.array_equal(st.mean, g["mean150"]) and np.array_equal(st.std, g["std150"])
    st.increment(list(rets[150:]), 150)
    assert np.array_equal(st.mean, g["mean300"]) and np.array_equal(st.std, g["std300"])
    js = st.to_json()
    assert np.array_equal(js["mean"], g["json_mean"]) and np.array_equal(js["var"], g["json_var"]) and js["count"] == g["json_count"]
    ob = WelfordRunningStat(5)
    ob.increment(g["obs"][:4], 4)
    ob.increment(g["obs"][4:], 8)
    assert np.array_equal(ob.mean, g["obs_mean"]) and np.array_equal(ob.std, g["obs_std"])
    a, b = WelfordRunningStat(5), WelfordRunningStat(5)
    a.increment(g["obs"][:5], 5)
    b.increment(g["obs"][5:], 7)
    a.increment_from_serialized_other(b.serialize())
    assert np.array_equal(a.running_mean, g["merged_mean"]) and np.array_equal(a.running_variance, g["merged_var"])
    assert a.count == g["merged_count"]
    c = WelfordRunningStat(1)
    c.from_json(js)
    assert c.count == st.count and np.allclose(c.std, st.std)


def test_host_exponential_certified_transform_equals_libm():
    """The vectorised transform of rlppo_torch_cpu_exponential accepts an element only when its float32 rounding is certain and
    sends the rest through libm (csrc/host_rng.cpp): over 2^22 values and several rates it must reproduce the libm-only path
    (rlppo_dbg_set(24, 0)) and torch itself bit for bit, with the same generator advance; ragged lengths cover the vector tails."""
    import ctypes
    import torch
    from rlgym_ppo_amd import _native as N
    L = N.lib()

    def draw(fast, seed, n, lam, threads):
        torch.manual_seed(seed)
        st = torch.get_rng_state().numpy().copy()
        out = np.empty(n, np.float32)
        assert L.rlppo_dbg_set(24, fast) == 0
        assert L.rlppo_torch_cpu_exponential(ctypes.c_void_p(st.ctypes.data), st.size, n, lam, ctypes.c_void_p(out.ctypes.data), threads) == 0
        return out, st

    try:
        for seed, n, lam, threads in ((1, 1 << 22, 1.0, 4), (2, 1000003, 1.0, 1), (3, 4099, 2.5, 2), (4, 7, 0.3, 1)):
            a, sa = draw(1, seed, n, lam, threads)
            b, sb = draw(0, seed, n, lam, threads)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(sa, sb), (seed, n, lam)
            torch.manual_seed(seed)
            ref = torch.empty(n).exponential_(lam)
            assert np.array_equal(a.view(np.uint32), ref.numpy().view(np.uint32)), (seed, n, lam)
            assert np.array_equal(sa, torch.get_rng_state().numpy())
    finally:
        L.rlppo_dbg_set(24, 1)


def test_host_exponential_is_torch_exponential_bit_for_bit():
    """rlppo_torch_cpu_exponential == torch.empty(n).exponential_(1) on the global CPU generator: same values, same generator
    advance (block boundaries of the MT19937 stream included), interleaved with other consumers of the generator; and the
    speculative look-ahead pipeline on top of it (engine.HostExponential) never changes the observable stream."""
    import ctypes
    import torch
    from rlgym_ppo_amd import _native as N
    from rlgym_ppo_amd import engine
    L = N.lib()

    def draw(n, threads):
        a = torch.get_rng_state().numpy().copy()
        out = np.empty(n, np.float32)
        N.check(L.rlppo_torch_cpu_exponential(ctypes.c_void_p(a.ctypes.data), a.size, n, 1.0, ctypes.c_void_p(out.ctypes.data), threads))
        torch.set_rng_state(torch.from_numpy(a))
        return torch.from_numpy(out)

    for seed, sizes in ((0, [1, 7, 90 * 4096, 3, 624, 625, 311, 90 * 64]), (123, [312, 312, 1, 4096 * 90, 10]), (5, [2 * 624 * 3, 5])):
        torch.manual_seed(seed)
        ref = [torch.empty(n).exponential_(1) for n in sizes]
        s_ref = torch.get_rng_state()
        torch.manual_seed(seed)
        got = [draw(n, 1 + i % 5) for i, n in enumerate(sizes)]
        assert all(torch.equal(a, b) for a, b in zip(ref, got)) and torch.equal(s_ref, torch.get_rng_state())
    torch.manual_seed(9)
    r = [torch.empty(1000).exponential_(1), torch.randn(33), torch.empty(5000).exponential_(1), torch.rand(7), torch.randint(0, 9, (5,))]
    torch.manual_seed(9)
    g = [draw(1000, 3), torch.randn(33), draw(5000, 2), torch.rand(7), torch.randint(0, 9, (5,))]
    assert all(torch.equal(a, b) for a, b in zip(r, g))
    assert L.rlppo_torch_cpu_exponential(None, 0, 4, 1.0, None, 1) != 0

    # the look-ahead pipeline: a rollout-like sequence with a foreign draw, a shape change and a re-seed in between
    def sequence(draw_):
        fn = lambda shape: draw_(shape).clone()         # the pipeline hands out views of recycled pinned buffers
        big = (768, 90)                                 # >= HostExponential.LOOKAHEAD_MIN elements: drawn ahead; small shapes on the spot
        torch.manual_seed(3)
        out = [fn(big), fn(big), fn(big)]
        out.append(torch.randn(5))                      # somebody else uses the generator: the speculation must be dropped
        out += [fn(big), fn((8 * 8, 3)), fn(big), fn(big), fn((64, 90)), fn((64, 90)), fn(big)]
        torch.manual_seed(4)                            # re-seeded
        out += [fn(big), fn(big)]
        return out, torch.get_rng_state()

    ref, s_ref = sequence(lambda shape: torch.empty(shape).exponential_(1))
    engine._HOST_EXP = None
    got, s_got = sequence(engine.host_exponential)
    assert all(torch.equal(a, b) for a, b in zip(ref, got)) and torch.equal(s_ref, s_got)
    assert engine._HOST_EXP.hits >= 3 and engine._HOST_EXP.misses >= 4   # both paths were exercised


def test_host_exponential_split_and_chained_forms_reproduce_torch():
    """[r3] The draw as two calls (stream phase / transform) and as chained one-call draws on two threads -- a call waits inside the
    library for its predecessor's stream phase, takes the state it left, publishes its own -- give torch's values and torch's
    generator states, draw after draw (engine.HostExponential pipelines consecutive rollout steps this way)."""
    import threading
    import torch
    from rlgym_ppo_amd import _native as N
    L = N.lib()
    P = lambda a: ctypes.c_void_p(a.ctypes.data)
    n, draws = 90 * 1500 + 7, 5
    torch.manual_seed(31)
    st0 = torch.get_rng_state().numpy().copy()
    want, states = [], []
    for _ in range(draws):
        want.append(torch.empty(n).exponential_(1).numpy())
        states.append(torch.get_rng_state().numpy().copy())
    # two calls
    st = st0.copy()
    words, out = np.empty(2 * n + 8, np.uint32), np.empty(n, np.float32)
    for k in range(draws):
        N.check(L.rlppo_torch_cpu_exponential_words(P(st), st.size, n, P(words)))
        N.check(L.rlppo_exponential_from_words(P(words), n, 1.0, P(out)))
        assert np.array_equal(out, want[k]) and np.array_equal(st, states[k])
    # chained: every draw on its own thread, started in REVERSE order so that every call really waits for its predecessor
    links = [np.zeros(N.EXP_LINK_HEADER + st0.size, np.uint8) for _ in range(draws)]
    outs = [np.empty(n, np.float32) for _ in range(draws)]
    scratch = [np.empty(2 * n + 8, np.uint32) for _ in range(draws)]
    rcs = [None] * draws

    def run(k):
        rcs[k] = L.rlppo_torch_cpu_exponential_chained(P(st0) if k == 0 else None, st0.size, n, 1.0, P(outs[k]), P(scratch[k]),
                                                       P(links[k - 1]) if k else None, P(links[k]))
    ths = [threading.Thread(target=run, args=(k,)) for k in reversed(range(draws))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert rcs == [0] * draws
    for k in range(draws):
        assert np.array_equal(outs[k], want[k]) and np.array_equal(links[k][N.EXP_LINK_HEADER:], states[k]) and links[k][0] == 1
    assert np.array_equal(st0, torch.manual_seed(31) and torch.get_rng_state().numpy())   # the start state was not modified
    # a failed predecessor is passed down the chain instead of hanging it
    bad = np.zeros(N.EXP_LINK_HEADER + st0.size, np.uint8)
    bad[:4] = np.frombuffer(np.int32(-1).tobytes(), np.uint8)
    nxt = np.zeros_like(bad)
    assert L.rlppo_torch_cpu_exponential_chained(None, st0.size, n, 1.0, P(outs[0]), P(scratch[0]), P(bad), P(nxt)) != 0
    assert np.frombuffer(nxt[:4].tobytes(), np.int32)[0] == -1
    assert L.rlppo_torch_cpu_exponential_chained(None, st0.size, n, 1.0, P(outs[0]), P(scratch[0]), None, P(nxt)) != 0   # no start state


def test_host_exponential_burst_prefetch_reproduces_torch_and_stays_transparent():
    """[r4] rlppo_torch_cpu_exponential_burst / HostExponential.prefetch: the next rollout's draws produced ahead in one go (two C
    calls walking the burst alternately).  (1) the C entry point alone: a burst of 7 draws over 2 and over 3 threads gives torch's
    values and torch's generator states, from a start state and from an earlier call's link block; a cancelled run marks its
    remaining draws failed.  (2) through HostExponential: draw, prefetch 11 more, 'learn' (no generator use), draw 12 -- values and
    the final generator state are torch's, every draw a hit; prefetch twice in a row is refused while the first is being served;
    (3) something else uses the generator between prefetch and the draws: the chain is dropped, the stream is still torch's."""
    import threading
    import torch
    from rlgym_ppo_amd import _native as N, engine
    L = N.lib()
    P = lambda a: ctypes.c_void_p(a.ctypes.data)
    n, draws = 90 * 1100 + 3, 7
    torch.manual_seed(77)
    st0 = torch.get_rng_state().numpy().copy()
    want, states = [], []
    for _ in range(draws):
        want.append(torch.empty(n).exponential_(1).numpy())
        states.append(torch.get_rng_state().numpy().copy())
    stride = N.EXP_LINK_HEADER + st0.size
    for threads in (2, 3):
        out, links, cancel = np.empty((draws, n), np.float32), np.zeros((draws, stride), np.uint8), np.zeros(1, np.int32)
        rcs = [None] * threads
        def run(j):
            rcs[j] = L.rlppo_torch_cpu_exponential_burst(P(st0), None, st0.size, n, 1.0, P(out), n, P(links), stride, j, threads, draws, P(cancel))
        ths = [threading.Thread(target=run, args=(j,)) for j in reversed(range(threads))]
        [t.start() for t in ths]
        [t.join() for t in ths]
        assert rcs == [0] * threads
        for k in range(draws):
            assert np.array_equal(out[k], want[k]) and np.array_equal(links[k][N.EXP_LINK_HEADER:], states[k])
            assert links[k].view(np.int32)[0] == 1 and links[k].view(np.int32)[1] == 1
    # continuing from an earlier call's link block: draws 3.. from the block draw 2 published
    out2, links2, cancel = np.empty((draws - 3, n), np.float32), np.zeros((draws - 3, stride), np.uint8), np.zeros(1, np.int32)
    assert L.rlppo_torch_cpu_exponential_burst(None, P(links[2]), st0.size, n, 1.0, P(out2), n, P(links2), stride, 0, 1, draws - 3, P(cancel)) == 0
    assert all(np.array_equal(out2[k], want[k + 3]) for k in range(draws - 3)) and np.array_equal(links2[-1][N.EXP_LINK_HEADER:], states[-1])
    # cancelled before it starts: every draw marked failed, nobody left waiting
    links3, cancel = np.zeros((draws, stride), np.uint8), np.ones(1, np.int32)
    assert L.rlppo_torch_cpu_exponential_burst(P(st0), None, st0.size, n, 1.0, P(out), n, P(links3), stride, 0, 1, draws, P(cancel)) != 0
    assert (links3.view(np.int32)[:, 1] == -1).all() and (links3.view(np.int32)[:, 0] == -1).all()
    assert L.rlppo_torch_cpu_exponential_burst(None, None, st0.size, n, 1.0, P(out), n, P(links3), stride, 0, 1, draws, P(cancel)) != 0  # no start

    # (2) through HostExponential
    shape = (1100, 90)
    torch.manual_seed(91)
    ref = [torch.empty(shape).exponential_(1) for _ in range(13)]
    s_ref = torch.get_rng_state()
    torch.manual_seed(91)
    h = engine.HostExponential()
    got = [h.draw(shape).clone()]
    added = h.prefetch(shape, 12)
    assert added == 12 - h.depth and len(h._chain) == 12
    assert h.prefetch(shape, 12) == 0 and h.prefetch(shape, 40) == 0        # already that long / storage busy: refused, nothing breaks
    got += [h.draw(shape).clone() for _ in range(12)]
    h._drain()
    assert all(torch.equal(a, b) for a, b in zip(ref, got)) and torch.equal(s_ref, torch.get_rng_state())
    assert h.hits == 12 and h.misses == 1
    # (3) an intruder between prefetch and the draws
    torch.manual_seed(92)
    ref = [torch.empty(shape).exponential_(1)]
    x_ref = torch.rand(5)
    ref += [torch.empty(shape).exponential_(1) for _ in range(4)]
    s_ref = torch.get_rng_state()
    torch.manual_seed(92)
    h = engine.HostExponential()
    got = [h.draw(shape).clone()]
    h.prefetch(shape, 9)
    x = torch.rand(5)                                                          # not ours: the prediction no longer holds
    got += [h.draw(shape).clone() for _ in range(4)]
    h._drain()
    assert torch.equal(x, x_ref) and all(torch.equal(a, b) for a, b in zip(ref, got)) and torch.equal(s_ref, torch.get_rng_state())
    assert h.misses >= 2 and all(not f.running() for b in h._burst.values() for f in b["futures"])


def test_host_exponential_survives_a_failing_look_ahead(monkeypatch):
    """A speculative draw that fails on its helper thread (here: every second one raises) must not change the observable stream:
    the request is drawn on the spot from the generator's real state, the chain is rebuilt, values and final state are torch's."""
    import torch
    from rlgym_ppo_amd import engine
    shape = (900, 90)   # above LOOKAHEAD_MIN: drawn ahead
    torch.manual_seed(44)
    want = [torch.empty(shape).exponential_(1) for _ in range(9)]
    s_want = torch.get_rng_state()
    h = engine.HostExponential()
    real, calls = h._chained_draw, [0]

    def flaky(e, prev_link):
        calls[0] += 1
        if calls[0] % 2 == 0:
            raise RuntimeError("injected failure")   # (before the library was even called: the guard must release the successor)
        return real(e, prev_link)
    monkeypatch.setattr(h, "_chained_draw", flaky)
    torch.manual_seed(44)
    got = [h.draw(shape).clone() for _ in range(9)]
    h._drain()
    assert all(torch.equal(a, b) for a, b in zip(want, got)) and torch.equal(s_want, torch.get_rng_state())
    assert h.hits >= 1 and h.misses >= 2 and calls[0] >= 6


def test_host_exponential_self_check_falls_back_to_torch(monkeypatch):
    """HostExponential verifies ONCE per process that librlppo's host exponential_ reproduces this torch build's stream (4096
    values and the generator state); a mismatch (another torch build) must not change the observable stream: torch's own
    exponential_ serves the draws, with a warning, and the generator ends where torch's would."""
    import warnings
    import torch
    from rlgym_ppo_amd import engine
    h = engine.HostExponential()
    torch.manual_seed(11)
    before = torch.get_rng_state()
    h._self_check()
    assert h.trusted and torch.equal(before, torch.get_rng_state())   # the check itself leaves the generator untouched

    bad = engine.HostExponential()
    real = bad._draw_into

    def skewed(state, buf, numel):   # a library whose stream differs in one value
        out = real(state, buf, numel)
        buf[0] += 1.0
        return out
    monkeypatch.setattr(bad, "_draw_into", skewed)
    torch.manual_seed(12)
    want = [torch.empty(64, 90).exponential_(1) for _ in range(3)]
    s_want = torch.get_rng_state()
    torch.manual_seed(12)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = [bad.draw((64, 90)).clone() for _ in range(3)]
    assert not bad.trusted and any("falling back" in str(x.message) for x in w)
    assert all(torch.equal(a, b) for a, b in zip(want, got)) and torch.equal(s_want, torch.get_rng_state())


def test_wait_words_tells_a_give_up_from_a_word_that_is_not_there_yet():
    """[r5] Host half of rlppo_act_opts.noise_ctl: rlppo_host_wait_words tells a completion word that carries the give-up bit (2)
    from one that is not there yet (1)."""
    from rlgym_ppo_amd import _native as N
    L = N.lib()
    words = np.array([5, 5, 0, 5], dtype=np.uint32)
    P = lambda a: ctypes.c_void_p(a.ctypes.data)
    assert L.rlppo_host_wait_words(P(words), 2, 5, 0) == 0 and L.rlppo_host_wait_words(P(words), 4, 5, 1000) == 1
    words[2] = 5 | 0x80000000
    assert L.rlppo_host_wait_words(P(words), 4, 5, 1000) == 2 and L.rlppo_host_wait_words(P(words), 2, 5, 0) == 0
    words[2] = 0x80000000                        # value 0x80000000 itself is just a value
    assert L.rlppo_host_wait_words(P(words[2:]), 1, 0x80000000, 0) == 0


def test_split_plane_image_is_conflict_free_for_ds_read_b128():
    """[r4] The 16-column block of a split-bf16 weight plane (csrc/gemm_split.hip, wpos(): the 16-byte chunk of column r, k quarter q
    sits at chunk 4 r + (q ^ (-(r // 4) & 3)) of the block's 1 KiB).  A fragment read is one ds_read_b128 with lane = 16 q + r, which
    the LDS serves in four groups of 16 lanes (MI355X_MICROARCH.md, LDS: {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32 each); a
    group is conflict-free when its 16 chunks fall on 16 different bank quads (chunk index mod 16).  The plain row image (chunk
    4 r + q) fails that test -- which is why the planes are packed this way; the tests of the GPU suite read them back through the
    same map."""
    wpos = lambda r, q: r * 4 + (q ^ ((-(r // 4)) & 3))
    plain = lambda r, q: r * 4 + q
    assert sorted(wpos(r, q) for r in range(16) for q in range(4)) == list(range(64))   # a permutation of the block's 64 chunks
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[lane + 32 for lane in g] for g in groups]
    assert sorted(lane for g in groups for lane in g) == list(range(64))
    for g in groups:
        quads = [wpos(lane % 16, lane // 16) % 16 for lane in g]
        assert len(set(quads)) == 16, quads
    assert any(len({plain(lane % 16, lane // 16) % 16 for lane in g}) < 16 for g in groups)
