"""GPU parity tests, kernel by kernel, through the C ABI (include/rlppo.h) against the CPU oracle / golden vectors.
Run on the MI355X box with:  python -m pytest tests -m gpu -x -q
Tolerances follow SURVEY.md section 8(c): indices exact (given identical probs + noise); forward probs/logp/values
rel 1e-5; GAE atol 1e-5 + rtol 1e-5; losses/grads rel 1e-5."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import gae as ogae  # noqa: E402
from oracle import nets, ppo  # noqa: E402
import fp64_gate  # noqa: E402  (tests/fp64_gate.py: the parity gate against float64 truth)


@pytest.fixture(scope="module")
def L():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from rlgym_ppo_amd import _native as N
    return N.lib()


def dev(x, dtype=torch.float32):
    return torch.as_tensor(np.asarray(x)).to("cuda", dtype=dtype).contiguous()


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def relerr(a, b):
    a = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, np.float64)
    b = np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def check(L, rc):
    assert rc == 0, L.rlppo_last_error()


# ------------------------------------------------------------------------------------------------ GEMMs
@pytest.mark.parametrize("M,N,K,epi", [
    (128, 128, 32, 0), (300, 256, 128, 1), (1000, 96, 256, 0), (77, 32, 256, 2), (513, 64, 64, 1), (256, 256, 96, 3),
    (4096, 256, 256, 3), (5, 128, 128, 1), (5000, 256, 128, 1), (3000, 96, 256, 0), (2048, 32, 256, 0), (1500, 256, 96, 3),
    (1100, 256, 32, 3), (70000, 256, 256, 1), (66000, 128, 64, 2), (1024, 64, 256, 1), (2051, 512, 512, 1), (999, 96, 512, 3),
])
def test_gemm_nt(L, M, N, K, epi):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K + 32, generator=g)          # lda > K
    B = torch.randn(N, K, generator=g) * (0.05 if epi == 2 else 1.0)  # keep tanh inputs O(1): fp32 input rounding
    bias = torch.randn(N, generator=g)
    mask = torch.randn(M, N, generator=g)
    Ad, Bd, bd, md = dev(A), dev(B), dev(bias), dev(mask)
    C = torch.full((M, N), float("nan"), device="cuda")
    check(L, L.rlppo_dbg_gemm_nt(stream(), P(Ad), A.shape[1], P(Bd), K, P(bd), P(md), N, P(C), N, M, N, K, epi))
    ref = A[:, :K].double() @ B.double().T
    if epi == 3:
        ref = ref * (mask > 0)
    else:
        ref = ref + bias.double()
        if epi == 1:
            ref = ref.clamp(min=0)
        if epi == 2:
            ref = torch.tanh(ref)
    assert relerr(C, ref) < (5e-6 if epi == 2 else 2e-6)


@pytest.mark.parametrize("M,N,K,hidden,epi", [(3072, 512, 512, 1, 1), (1000, 256, 256, 1, 1), (4096 + 77, 512, 256, 1, 1), (700, 128, 64, 1, 1),
                                              (3000, 32, 512, 0, 2), (2500, 96, 256, 0, 0), (513, 256, 128, 0, 0)])
def test_gemm_nt_b16(L, M, N, K, hidden, epi, b16_tiles):
    """The bf16-in-memory forward product (bf16 update precision): exact bf16 products, fp32 accumulation.  Hidden form: the
    three outputs (bf16, the same values as fp32, ReLU bitmask) are mutually consistent and equal the float64 product rounded to
    bf16 except where fp32 summation noise crosses a rounding boundary; the bitmask is the one the fp32 dX kernel expects."""
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g)).bfloat16()
    W = (torch.randn(N, K, generator=g) * (0.05 if epi == 2 else 0.1)).bfloat16()
    bias = torch.randn(N, generator=g) * 0.1
    Ad, Wd, bd = A.cuda(), W.cuda(), bias.cuda()
    C = torch.full((M, N), float("nan"), device="cuda")
    ref = A.double() @ W.double().T + bias.double()
    if not hidden:
        check(L, L.rlppo_dbg_gemm_nt_b16(stream(), P(Ad), K, P(Wd), K, P(bd), P(C), N, None, 0, M, N, K, epi, 0, None))
        ref = torch.tanh(ref) if epi == 2 else ref
        assert relerr(C, ref) < 5e-6
        return
    Cb = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    bits = torch.zeros(int(L.rlppo_dbg_gemm_nt_bits_bytes(M, N)), dtype=torch.uint8, device="cuda")
    check(L, L.rlppo_dbg_gemm_nt_b16(stream(), P(Ad), K, P(Wd), K, P(bd), P(C), N, P(Cb), N, M, N, K, 1, 1, P(bits)))
    torch.cuda.synchronize()
    assert torch.equal(C.cpu(), Cb.float().cpu())                       # the fp32 copy IS the bf16 value
    want = ref.clamp(min=0).float().bfloat16().float()
    diff = (C.cpu() - want).abs()
    # at most one bf16 ulp (+ the fp32 summation noise of a K-term sum, which dominates where the sum nearly cancels), and
    # only where that noise crosses a rounding boundary
    ulp = want.abs() * 2.0 ** -7 + 2.0 ** -24 * K * (A.float().abs() @ W.float().abs().T + bias.abs()).double()
    assert (diff <= ulp).all() and (diff > 0).float().mean().item() < 2e-3
    # the bitmask drives the fp32 masked-dX kernel: dX = (dY . Wt^T) masked by it must equal masking by C > 0
    dY = torch.randn(M, 128, device="cuda")
    Wt = torch.randn(N, 128, device="cuda") * 0.05
    D0, D1 = torch.empty(M, N, device="cuda"), torch.full((M, N), -7.0, device="cuda")
    check(L, L.rlppo_dbg_gemm_nt(stream(), P(dY), 128, P(Wt), 128, None, P(C), N, P(D0), N, M, N, 128, 3))
    check(L, L.rlppo_dbg_gemm_nt_bits(stream(), P(dY), 128, P(Wt), 128, None, P(D1), N, M, N, 128, 3, P(bits)))
    torch.cuda.synchronize()
    assert torch.equal(D0, D1)


@pytest.fixture(params=[2, 1, 0], ids=["tile256_persistent", "tile256", "tile128"])
def b16_tiles(L, request):
    """The forms of the bf16 hidden / dX / dW products (rlppo_dbg_set(23)): 256 x 256 tiles walked by persistent workgroups (default
    where it applies [r3]), 256 x 256 tiles with one workgroup per tile, 128 x 128 tiles."""
    check(L, L.rlppo_dbg_set(23, request.param))
    yield request.param
    check(L, L.rlppo_dbg_set(23, 2))


@pytest.mark.parametrize("M,N,K", [(3072, 512, 512), (4096 + 77, 256, 512), (700, 128, 64), (65536, 512, 512)])
def test_gemm_nt_b16_dx(L, M, N, K, b16_tiles):
    """The backward product of the bf16 update precision: dX = round_bf16(dY . W^T-operand) masked by the ReLU bitmask the hidden
    forward of the same M x N geometry wrote.  bf16 in, bf16 out; ragged last row tile; the mask is exact."""
    g = torch.Generator().manual_seed(M + N + K + 1)
    # a hidden forward of width N produces the bitmask (and the activation it describes)
    A0 = torch.randn(M, 64, generator=g).bfloat16().cuda()
    W0 = (torch.randn(N, 64, generator=g) * 0.1).bfloat16().cuda()
    b0 = (torch.randn(N, generator=g) * 0.1).cuda()
    H = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    bits = torch.zeros(int(L.rlppo_dbg_gemm_nt_bits_bytes(M, N)), dtype=torch.uint8, device="cuda")
    check(L, L.rlppo_dbg_gemm_nt_b16(stream(), P(A0), 64, P(W0), 64, P(b0), None, 0, P(H), N, M, N, 64, 1, 1, P(bits)))
    dY = torch.randn(M, K, generator=g).bfloat16()
    Wt = (torch.randn(N, K, generator=g) * 0.05).bfloat16()       # W^T[pin = N][pout = K]
    dYd, Wtd = dY.cuda(), Wt.cuda()
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    check(L, L.rlppo_dbg_gemm_nt_b16(stream(), P(dYd), K, P(Wtd), K, None, None, 0, P(out), N, M, N, K, 3, 2, P(bits)))
    torch.cuda.synchronize()
    on = (H > 0).cpu()
    assert 0.3 < on.float().mean().item() < 0.7
    got = out.float().cpu()
    assert (got[~on] == 0).all() and not torch.isnan(got).any()
    prod = dY.double() @ Wt.double().T
    want = prod.float().bfloat16().float() * on
    diff = (got - want).abs()
    ulp = want.abs() * 2.0 ** -7 + 2.0 ** -24 * K * (dY.float().abs() @ Wt.float().abs().T).double()
    assert (diff <= ulp).all() and (diff > 0).float().mean().item() < 2e-3
    assert L.rlppo_dbg_gemm_nt_b16(stream(), P(dYd), K, P(Wtd), K, None, None, 0, P(out), N, M, N, K, 3, 2, None) != 0  # no bitmask


@pytest.mark.parametrize("M,pout,pin,out,in_", [(4096, 128, 128, 128, 128), (1000, 256, 128, 256, 107), (70000, 512, 512, 512, 512),
                                               (33, 128, 256, 100, 231), (65536, 512, 256, 512, 231), (5000 + 13, 256, 384, 256, 384),
                                               (3000 + 7, 256, 256, 200, 256), (100, 512, 256, 512, 256)])
def test_gemm_tn_b16(L, M, pout, pin, out, in_, b16_tiles):
    """The weight-gradient product of the bf16 update precision: dW += dY^T . X, db += colsum(dY), both operands bf16 in memory,
    contraction over rows through the transposing LDS read; partial tiles + the fixed-order reduction of the fp32 form: ragged last
    stage, padded columns, accumulation on top of existing gradients, bit-identical from run to run."""
    g = torch.Generator().manual_seed(M + pout + in_)
    dY = torch.zeros(M, pout)
    dY[:, :out] = torch.randn(M, out, generator=g)
    X = torch.zeros(M, pin + 8)                      # ldx > pin
    X[:, :in_] = torch.randn(M, in_, generator=g)
    dYb, Xb = dY.bfloat16(), X.bfloat16()
    dW0, db0 = torch.randn(out, in_, generator=g), torch.randn(out, generator=g)
    dYd, Xd = dYb.cuda(), Xb.cuda()
    ws = torch.empty(int(L.rlppo_dbg_gemm_tn_workspace_bytes(out, in_, M)), dtype=torch.uint8, device="cuda")
    ws.fill_(0xFF)
    results = []
    for _ in range(3):
        dW, db = dev(dW0), dev(db0)
        check(L, L.rlppo_dbg_gemm_tn_b16(stream(), P(dYd), pout, P(Xd), pin + 8, P(dW), P(db), pout, pin, out, in_, M, P(ws), ws.numel()))
        results.append(dW.clone())
    refW = dW0.double() + dYb[:, :out].double().T @ Xb[:, :in_].double()
    refb = db0.double() + dYb[:, :out].double().sum(0)
    assert relerr(results[0], refW) < 2e-6 and relerr(db, refb) < 2e-6
    assert torch.equal(results[0], results[1]) and torch.equal(results[0], results[2])
    assert L.rlppo_dbg_gemm_tn_b16(stream(), P(dYd), pout, P(Xd), pin + 8, P(dW), P(db), pout, pin + 32, out, in_, M, P(ws), ws.numel()) != 0


@pytest.mark.parametrize("M,out,kp,in_", [(3000, 1, 512, 512), (4096 + 50, 16, 512, 512), (777, 21, 128, 100), (70000, 8, 256, 256),
                                         (600, 32, 1024, 1024)])
def test_thin_head_b16(L, M, out, kp, in_):
    """Narrow output layers (<= 32 outputs) in the bf16 update precision: dX = round_bf16(dY . W) masked by the hidden layer's ReLU
    bitmask (bf16 out), dW += dY^T . hb, db += colsum(dY) with hb the bf16 activation; ragged row tiles, accumulation on top of
    existing gradients, bit-identical from run to run."""
    g = torch.Generator().manual_seed(M + out + kp)
    A0 = torch.randn(M, 64, generator=g).bfloat16().cuda()
    W0 = (torch.randn(kp, 64, generator=g) * 0.1).bfloat16().cuda()
    b0 = (torch.randn(kp, generator=g) * 0.1).cuda()
    hb = torch.zeros(M, kp, dtype=torch.bfloat16, device="cuda")                 # the hidden activation and its bitmask
    bits = torch.zeros(int(L.rlppo_dbg_gemm_nt_bits_bytes(M, kp)), dtype=torch.uint8, device="cuda")
    check(L, L.rlppo_dbg_gemm_nt_b16(stream(), P(A0), 64, P(W0), 64, P(b0), None, 0, P(hb), kp, M, kp, 64, 1, 1, P(bits)))
    dY = torch.zeros(M, 32)
    dY[:, :out] = torch.randn(M, out, generator=g)
    W = torch.zeros(32, kp)
    W[:out] = (torch.randn(out, kp, generator=g) * 0.05).bfloat16().float()
    dW0, db0 = torch.randn(out, in_, generator=g), torch.randn(out, generator=g)
    dYd, Wd = dev(dY), dev(W)
    ws = torch.empty(int(L.rlppo_dbg_thin_head_workspace_bytes(out, kp, M)), dtype=torch.uint8, device="cuda")
    ws.fill_(0xFF)
    res = []
    for _ in range(2):
        dxb = torch.full((M, kp), float("nan"), dtype=torch.bfloat16, device="cuda")
        dW, db = dev(dW0), dev(db0)
        check(L, L.rlppo_dbg_thin_head_b16(stream(), P(dYd), 32, out, P(Wd), kp, P(bits), P(hb), kp, P(dxb), P(dW), P(db), in_, kp, M,
                                           P(ws), ws.numel()))
        res.append((dxb.clone(), dW.clone(), db.clone()))
    torch.cuda.synchronize()
    on = (hb > 0).cpu()
    h = hb.float().cpu()
    got = res[0][0].float().cpu()
    prod = dY[:, :out].double() @ W[:out].double()
    want = prod.float().bfloat16().float() * on
    diff = (got - want).abs()
    ulp = want.abs() * 2.0 ** -7 + 2.0 ** -22 * (dY[:, :out].abs() @ W[:out].abs()).double()
    assert (got[~on] == 0).all() and (diff <= ulp).all() and (diff > 0).float().mean().item() < 2e-3
    refW = dW0.double() + dY[:, :out].double().T @ h[:, :in_].double()
    refb = db0.double() + dY[:, :out].double().sum(0)
    assert relerr(res[0][1], refW) < 2e-6 and relerr(res[0][2], refb) < 2e-6
    assert all(torch.equal(a, b) for a, b in zip(res[0], res[1]))


def test_relu_bitmask_forms_are_bitwise_equal(L):
    """The hidden-layer forward that also writes the ReLU bitmask gives the same activations as the plain
    forward, and the dX product masked by that bitmask gives the same result as the one masked by re-reading the activation
    (ragged last row tile, exact zeros and negative pre-activations in the mask)."""
    torch.manual_seed(6)
    M = 4096 * 3 + 77
    A = torch.randn(M, 256, device="cuda")
    W = torch.randn(256, 256, device="cuda") * 0.05
    b = torch.randn(256, device="cuda") * 0.1
    dY = torch.randn(M, 128, device="cuda")
    Wt = torch.randn(256, 128, device="cuda") * 0.05
    bits = torch.zeros(int(L.rlppo_dbg_gemm_nt_bits_bytes(M, 256)), dtype=torch.uint8, device="cuda")
    assert bits.numel() == ((M + 127) // 128) * 2 * 256 * 8
    H0 = torch.empty(M, 256, device="cuda")
    H1 = torch.full((M, 256), -7.0, device="cuda")
    check(L, L.rlppo_dbg_gemm_nt(stream(), P(A), 256, P(W), 256, P(b), None, 0, P(H0), 256, M, 256, 256, 1))
    check(L, L.rlppo_dbg_gemm_nt_bits(stream(), P(A), 256, P(W), 256, P(b), P(H1), 256, M, 256, 256, 1, P(bits)))
    assert torch.equal(H0, H1) and (H0 == 0).float().mean().item() > 0.3
    D0 = torch.empty(M, 256, device="cuda")
    D1 = torch.full((M, 256), -7.0, device="cuda")
    check(L, L.rlppo_dbg_gemm_nt(stream(), P(dY), 128, P(Wt), 128, None, P(H0), 256, P(D0), 256, M, 256, 128, 3))
    check(L, L.rlppo_dbg_gemm_nt_bits(stream(), P(dY), 128, P(Wt), 128, None, P(D1), 256, M, 256, 128, 3, P(bits)))
    torch.cuda.synchronize()
    assert torch.equal(D0, D1)
    assert ((D1 == 0) == (H0 == 0)).all()      # zero exactly where the unit was off (N(0,1) products are never exactly 0)
    assert L.rlppo_dbg_gemm_nt_bits(stream(), P(A), 256, P(W), 256, P(b), P(H1), 96, M, 96, 256, 1, P(bits)) != 0   # width not 128 k


@pytest.mark.parametrize("M,out,in_", [(1000, 90, 256), (4096, 256, 107), (70000, 256, 256), (33, 1, 64), (5000, 300, 130),
                                       (1000, 256, 107), (33, 90, 256), (2500, 1, 256), (1, 21, 32), (3000, 512, 231)])
def test_gemm_tn(L, M, out, in_):
    """dW += dY^T . X, db += colsum(dY) through partial 128x128 tiles per split + a fixed-order reduction kernel (no fp32
    atomics): ragged last stage / tile included, accumulation on top of existing gradients, bit-identical from run to run."""
    g = torch.Generator().manual_seed(M + out)
    ny, kx = int(L.rlppo_padded_out(out)), int(L.rlppo_padded_out(in_))
    dY = torch.zeros(M, ny)
    dY[:, :out] = torch.randn(M, out, generator=g)
    X = torch.zeros(M, kx)
    X[:, :in_] = torch.randn(M, in_, generator=g)
    dW0, db0 = torch.randn(out, in_, generator=g), torch.randn(out, generator=g)
    dYd, Xd = dev(dY), dev(X)
    ws = torch.empty(int(L.rlppo_dbg_gemm_tn_workspace_bytes(out, in_, M)), dtype=torch.uint8, device="cuda")
    assert ws.numel() > 0
    ws.fill_(0xFF)  # NaN patterns: slots the kernel does not write must not reach the sums
    results = []
    for _ in range(3):
        dW, db = dev(dW0), dev(db0)
        check(L, L.rlppo_dbg_gemm_tn(stream(), P(dYd), ny, ny, P(Xd), kx, kx, P(dW), P(db), out, in_, M, P(ws), ws.numel()))
        results.append(dW.clone())
    refW = dW0.double() + dY[:, :out].double().T @ X[:, :in_].double()
    refb = db0.double() + dY[:, :out].double().sum(0)
    assert relerr(results[0], refW) < 2e-6 and relerr(db, refb) < 2e-6
    assert torch.equal(results[0], results[1]) and torch.equal(results[0], results[2])


_TN_GROUP_DATA = {}   # the host data and float64 products of the LAST _tn_group_case key (a test calls it three times with the same one)


def _tn_group_case(L, M, shapes, seed, gather_first=False, budget=0, reverse=False):
    """dW / db of several products through ONE rlppo_dbg_gemm_tn_group call; returns ([(dW, db)], [(refW, refb)] in float64).
    reverse: the same products (same data) handed over in reverse order."""
    from rlgym_ppo_amd import _native as N
    key = (M, tuple(shapes), seed, gather_first)
    if key not in _TN_GROUP_DATA:
        _TN_GROUP_DATA.clear()
        g = torch.Generator().manual_seed(seed)
        n_src = 3 * M + 7
        rowtab = torch.randint(0, n_src, (M,), generator=g).to(torch.int32)
        host, refs = [], []
        for i, (out, in_) in enumerate(shapes):
            ny, kx = int(L.rlppo_padded_out(out)), int(L.rlppo_padded_out(in_))
            dY = torch.zeros(M, ny)
            dY[:, :out] = torch.randn(M, out, generator=g)
            gathered = gather_first and i == 0
            rows = n_src if gathered else M
            X = torch.zeros(rows, kx)
            X[:, :in_] = torch.randn(rows, in_, generator=g)
            dW0, db0 = torch.randn(out, in_, generator=g), torch.randn(out, generator=g)
            host.append((dY, X, dW0, db0))
            # float64 truth on the GPU (torch's float64 product of the same float32 values: the yardstick, 1e-16; the CPU took 10 s at 300,000 rows)
            Xr = dev(X)[rowtab.long().cuda()] if gathered else dev(X)
            refs.append(((dev(dW0).double() + dev(dY)[:, :out].double().T @ Xr[:, :in_].double()).cpu(),
                         (dev(db0).double() + dev(dY)[:, :out].double().sum(0)).cpu()))
            del Xr
        _TN_GROUP_DATA[key] = (rowtab, host, refs)
    rowtab, host, refs = _TN_GROUP_DATA[key]
    n_src = 3 * M + 7
    prods = (N.TnProduct * len(shapes))()
    keep, outs = [], []
    for i, (out, in_) in enumerate(shapes):
        ny, kx = int(L.rlppo_padded_out(out)), int(L.rlppo_padded_out(in_))
        dY, X, dW0, db0 = host[i]
        gathered = gather_first and i == 0
        rows = n_src if gathered else M
        dYd, Xd, dW, db = dev(dY), dev(X), dev(dW0), dev(db0)
        rt = rowtab.cuda() if gathered else None
        keep.append((dYd, Xd, rt))
        q = prods[len(shapes) - 1 - i if reverse else i]
        q.dY, q.ldy, q.ny_valid, q.X, q.ldx, q.kx_valid = dYd.data_ptr(), ny, ny, Xd.data_ptr(), kx, kx
        q.dW, q.db, q.out, q.in_ = dW.data_ptr(), db.data_ptr(), out, in_
        q.rowtab, q.src_rows = (rt.data_ptr(), rows) if gathered else (None, 0)
        outs.append((dW, db))
    check(L, L.rlppo_dbg_set(38, budget))
    try:
        ws = torch.empty(int(L.rlppo_dbg_gemm_tn_group_workspace_bytes(prods, len(shapes), M)), dtype=torch.uint8, device="cuda")
        assert ws.numel() > 0
        ws.fill_(0xFF)  # NaN patterns: slots the kernels do not write must not reach the sums
        check(L, L.rlppo_dbg_gemm_tn_group(stream(), prods, len(shapes), M, P(ws), ws.numel()))
        torch.cuda.synchronize()
    finally:
        check(L, L.rlppo_dbg_set(38, 0))
    return outs, refs


@pytest.mark.parametrize("M,shapes,gather_first,budget", [
    (65536, [(256, 107), (256, 256), (256, 256), (90, 256), (256, 107), (256, 256), (256, 256)], True, 0),   # the update's own set (cfg2, one rank of 8)
    (1500, [(256, 107), (256, 256), (90, 256)], True, 0),          # ragged: partial stages, fewer rows than a full split
    (33, [(90, 256), (21, 32), (300, 130)], False, 0),             # hardly any rows: one or two splits per product
    (5000, [(512, 231), (16, 512), (256, 256)], False, 0),
    (70001, [(256, 256), (90, 256)], False, 24),                   # a small grid: many rows per split, splits that are not multiples of 8
    (4096, [(256, 256)] * 20, False, 0),                           # many products: the kernel-argument table well filled
    (300000, [(256, 256), (256, 107), (90, 256)], True, 0),        # more rows than one round of 8192-row splits holds: several rounds
])
def test_gemm_tn_group(L, M, shapes, gather_first, budget):
    """[r5] Every weight-gradient product of a pass in ONE launch + ONE fixed-order reduction (rlppo_dbg_gemm_tn_group, the form
    rlppo_ppo_minibatch uses): each product against float64, accumulated on top of existing gradients, the first one through a row
    table (the fused minibatch gather); bit-identical from run to run and whatever the ORDER of the products in the call."""
    outs, refs = _tn_group_case(L, M, shapes, seed=M + len(shapes), gather_first=gather_first, budget=budget)
    # (a forced small grid makes every split one fp32 chain of ~17,500 rows: its rounding grows with the square root of that length;
    # the library's own plans keep a split within 8192 rows)
    tol = 1e-5 if budget else 2e-6
    for (dW, db), (rW, rb) in zip(outs, refs):
        assert relerr(dW, rW) < tol and relerr(db, rb) < tol
    again, _ = _tn_group_case(L, M, shapes, seed=M + len(shapes), gather_first=gather_first, budget=budget)
    for (a, b), (c, e) in zip(outs, again):
        assert torch.equal(a, c) and torch.equal(b, e)
    if len(shapes) <= 16:  # the plan depends on the SET of shapes: the same products in reverse order, the same bits
        rev, _ = _tn_group_case(L, M, shapes, seed=M + len(shapes), gather_first=gather_first, budget=budget, reverse=True)
        for (a, b), (c, e) in zip(outs, rev):
            assert torch.equal(a, c) and torch.equal(b, e)


# -------------------------------------------------------------------------------------------------- GAE
def run_gae(L, rews, dones, trunc, values, gamma, lmbda, std):
    n = len(rews)
    vt, adv, ret = (torch.empty(n, device="cuda") for _ in range(3))
    ws = torch.empty(int(L.rlppo_gae_workspace_bytes(n)), dtype=torch.uint8, device="cuda")
    r, d, t, v = dev(rews), dev(dones), dev(np.asarray(trunc, np.float32)), dev(values)
    check(L, L.rlppo_gae(stream(), P(r), P(d), P(t), P(v), n,
                         gamma, lmbda, float("nan") if std is None else float(std), P(vt), P(adv), P(ret), P(ws), ws.numel()))
    return vt.cpu().numpy(), adv.cpu().numpy(), ret.cpu().numpy()


GAE_DEFAULT_FORM = 1


@pytest.fixture(params=[1, 0], ids=["lookback", "two_launch"], autouse=False)
def gae_algo(L, request):
    """Both GAE implementations are held to the same checks: single-pass decoupled look-back (default) and the
    two-launch summary + apply form (what a stream capture gets)."""
    check(L, L.rlppo_dbg_set(1, request.param))
    yield request.param
    check(L, L.rlppo_dbg_set(1, GAE_DEFAULT_FORM))


def test_gae_lookback_stress_repeated(L):
    # the look-back hand-off is timing dependent: repeat a many-chunk scan and demand bit-identical outputs every time
    rews, dones, trunc, values = synth_gae(8192, 256, seed=5, p_mid=0.0)
    dones[:] = 0
    trunc[:] = 0           # no trajectory ends at all: every chunk must chain through ALL chunks to its right
    trunc[-1] = 1
    ref = run_gae(L, rews, dones, trunc, values, 0.999, 0.999, None)
    ovt, oadv, oret = ogae.gae(rews, dones, trunc, values, 0.999, 0.999, None, "f64")
    np.testing.assert_allclose(ref[1], oadv, rtol=1e-5, atol=1e-4)
    for _ in range(20):
        again = run_gae(L, rews, dones, trunc, values, 0.999, 0.999, None)
        for a, b in zip(ref, again):
            assert np.array_equal(a, b)


def test_gae_forms_agree_on_ragged_sizes(L):
    """The single-pass and the two-launch form compose the same float64 maps in different association orders; both round once at
    the store.  They agree to the last float32 bit or one ulp of it, on sizes that exercise the ragged tail of every lane / wave /
    chunk position (n mod 4, mod 8, mod 512, mod 2048 all non-zero: full waves leave through the LDS transpose, a ragged wave
    element by element) and the general look-back path (no trajectory end for long runs)."""
    for n, p_mid in ((1, 0.0), (3, 0.0), (255, 0.01), (257, 0.0), (515, 0.02), (1030, 0.0), (2047, 0.0), (2049, 0.0), (4099, 0.001), (70001, 0.0005), (300007, 0.0)):
        rs = np.random.RandomState(n)
        rews, values = rs.randn(n).astype(np.float32), rs.randn(n + 1).astype(np.float32)
        dones = (rs.rand(n) < p_mid).astype(np.float32)
        trunc = ((rs.rand(n) < p_mid) & (dones == 0)).astype(np.float32)
        outs = {}
        for form in (1, 0):
            check(L, L.rlppo_dbg_set(1, form))
            outs[form] = run_gae(L, rews, dones, trunc, values, 0.995, 0.97, 1.3)
        check(L, L.rlppo_dbg_set(1, GAE_DEFAULT_FORM))
        o = ogae.gae(rews, dones, trunc, values, 0.995, 0.97, 1.3, "f64")
        for k in range(3):
            np.testing.assert_allclose(outs[1][k], np.asarray(o[k], np.float32), rtol=2e-6, atol=2e-6, err_msg=f"n={n} out {k}")
            np.testing.assert_allclose(outs[1][k], outs[0][k], rtol=3e-7, atol=1e-6, err_msg=f"n={n} out {k}")


def test_gae_golden_vectors(L, golden, gae_algo):
    g = golden("g3_gae")
    for c in range(int(g["n_cases"])):
        p = f"c{c}."
        std = g[p + "ret_std"]
        std = None if np.isnan(std) else std
        gamma = float(g[p + "gamma"]) if p + "gamma" in g else 0.99
        lm = float(g[p + "lmbda"]) if p + "lmbda" in g else 0.95
        vt, adv, ret = run_gae(L, g[p + "rews"], g[p + "dones"], g[p + "trunc"], g[p + "values"], gamma, lm, std)
        np.testing.assert_allclose(vt, g[p + "value_targets"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(adv, g[p + "advantages"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(ret, g[p + "returns"], rtol=1e-5, atol=1e-5)


def synth_gae(n_seg, seg_len, seed=0, p_mid=0.005):
    """trajectory-major segments, end = done|truncated (p 0.5), iid mid-segment dones."""
    rs = np.random.RandomState(seed)
    n = n_seg * seg_len
    rews = rs.randn(n).astype(np.float32)
    values = rs.randn(n + 1).astype(np.float32)
    dones = (rs.rand(n) < p_mid).astype(np.float32)
    trunc = np.zeros(n, np.float32)
    ends = np.arange(seg_len - 1, n, seg_len)
    is_done = rs.rand(n_seg) < 0.5
    dones[ends[is_done]] = 1
    dones[ends[~is_done]] = 0
    trunc[ends[~is_done]] = 1
    return rews, dones, trunc, values


@pytest.mark.parametrize("n_seg,seg_len,std", [(64, 256, 1.7), (1000, 37, None), (3, 5000, 0.01), (1, 2049, 1.0), (8192, 256, 1.7)])
def test_gae_matches_oracle(L, n_seg, seg_len, std, gae_algo):
    rews, dones, trunc, values = synth_gae(n_seg, seg_len, seed=n_seg)
    vt, adv, ret = run_gae(L, rews, dones, trunc, values, 0.99, 0.95, std)
    ovt, oadv, oret = ogae.gae(rews, dones, trunc, values, 0.99, 0.95, std, "f64")
    # float64 scan vs float64 sequential: agreement is at the level of the final fp32 rounding
    np.testing.assert_allclose(adv, oadv, rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(vt, ovt, rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(ret, oret.astype(np.float32), rtol=2e-6, atol=2e-6)


def test_gae_long_undiscounted_carry(L, gae_algo):
    # no episode end for 300k steps and gamma = lambda = 1: the carry must cross ~146 workgroups exactly
    rs = np.random.RandomState(1)
    n = 300_001
    rews = (rs.randn(n) * 0.01).astype(np.float32)
    values = rs.randn(n + 1).astype(np.float32)
    z = np.zeros(n, np.float32)
    vt, adv, ret = run_gae(L, rews, z, z, values, 1.0, 1.0, None)
    ovt, oadv, oret = ogae.gae(rews, z, z, values, 1.0, 1.0, None, "f64")
    np.testing.assert_allclose(ret, oret, rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(adv, oadv, rtol=1e-6, atol=1e-5)


def test_gae_segment_independence_and_linearity(L, gae_algo):
    # size-independent properties at the BASELINE size (8192 x 256): (1) outputs before a done do not depend on what
    # follows it; (2) returns are linear in the rewards.
    rews, dones, trunc, values = synth_gae(8192, 256, seed=0)
    vt, adv, ret = run_gae(L, rews, dones, trunc, values, 0.99, 0.95, None)
    cut = 1_000_000
    end = cut + int(np.argmax((dones[cut:] + trunc[cut:]) > 0))  # first trajectory end at/after cut
    r2, v2 = rews.copy(), values.copy()
    r2[end + 1:] = 7.0
    v2[end + 2:] = -3.0
    vt2, adv2, ret2 = run_gae(L, r2, dones, trunc, v2, 0.99, 0.95, None)
    if dones[end] == 1:
        assert np.array_equal(adv[:end + 1], adv2[:end + 1])
    assert np.array_equal(ret[:end + 1], ret2[:end + 1])
    _, _, ret3 = run_gae(L, 2 * rews, dones, trunc, values, 0.99, 0.95, None)
    np.testing.assert_allclose(ret3, 2 * ret, rtol=1e-6, atol=1e-6)


def test_gae_recycled_workspace_and_loud_timeout(L):
    """Tags are per launch (a kernel argument from a process-wide counter), so a workspace full of another launch's records or of
    garbage does not matter, and launches never race on an epoch word; the header's timeout counter stays 0 in normal runs.
    A look-back wait that does time out (forced here with a spin limit of 0 on a scan with no trajectory end at all: every chunk
    must chain) never passes silently: the affected outputs are NaN and the counter says how many waits gave up."""
    rews, dones, trunc, values = synth_gae(96, 256, seed=9, p_mid=0.0)
    dones[:] = 0
    trunc[:] = 0
    trunc[-1] = 1
    n = len(rews)
    ovt, oadv, oret = ogae.gae(rews, dones, trunc, values, 0.99, 0.95, 1.3, "f64")
    vt, adv, ret = (torch.empty(n, device="cuda") for _ in range(3))
    ws = torch.empty(int(L.rlppo_gae_workspace_bytes(n)), dtype=torch.uint8, device="cuda")
    r, d, t, v = dev(rews), dev(dones), dev(trunc), dev(values)
    call = lambda: check(L, L.rlppo_gae(stream(), P(r), P(d), P(t), P(v), n, 0.99, 0.95, 1.3, P(vt), P(adv), P(ret), P(ws), ws.numel()))
    for fill in (0x00, 0xFF, 0x5A):
        ws.fill_(fill)
        ws[:16] = 0
        call()
        call()  # second launch on a workspace that holds the first launch's records
        np.testing.assert_allclose(adv.cpu().numpy(), oadv, rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(ret.cpu().numpy(), oret.astype(np.float32), rtol=2e-6, atol=2e-6)
        assert int(ws[4:8].view(torch.int32).item()) == 0, "a look-back wait timed out in a normal run"
    check(L, L.rlppo_dbg_set(21, 0))
    try:
        ws[:16] = 0
        call()
        timeouts = int(ws[4:8].view(torch.int32).item())
        a = adv.cpu().numpy()
        assert timeouts > 0 and np.isnan(a).any()
        ok = ~np.isnan(a)                                   # what is not poisoned is right (the rightmost chunk at least)
        assert ok[-2048:].all()
        np.testing.assert_allclose(a[ok], oadv[ok], rtol=2e-6, atol=2e-6)
    finally:
        check(L, L.rlppo_dbg_set(21, -1))


def test_gae_timeout_raises_through_every_python_caller(L):
    """A timed-out look-back wait in a chunk k > 0 with a trajectory end to its left leaves the HEAD of the outputs clean, so a
    caller that looks at returns[:150] alone (round 2's only guard) would train on NaN advantages.  gae_device clears the header's
    counter before the launch and reads it back after: the failure raises for every caller, whatever it looks at."""
    from rlgym_ppo_amd.util import torch_functions as TF
    rews, dones, trunc, values = synth_gae(96, 256, seed=9, p_mid=0.0)
    dones[:] = 0
    trunc[:] = 0
    trunc[-1] = 1
    dones[2048 + 100] = 1     # a trajectory end at the start of chunk 1: chunk 0 resolves its carry from its own look-ahead
    args = (dev(rews), dev(dones), dev(trunc), dev(values), 0.99, 0.95, 1.3)
    vt, adv, ret = TF.gae_device(*args)   # normal run: no exception, outputs right
    np.testing.assert_allclose(adv.cpu().numpy(), ogae.gae(rews, dones, trunc, values, 0.99, 0.95, 1.3, "f64")[1], rtol=2e-6, atol=2e-6)
    check(L, L.rlppo_dbg_set(21, 0))
    try:
        _, _, ret = TF.gae_device(*args, check=False)
        r = ret.cpu().numpy()
        assert not np.isnan(r[:150]).any() and np.isnan(r).any()   # exactly the case the old guard missed
        with pytest.raises(TF.GAETimeout):
            TF.gae_device(*args)
        with pytest.raises(TF.GAETimeout):
            TF.compute_gae(rews, dones, trunc, values, 0.99, 0.95, 1.3)
    finally:
        check(L, L.rlppo_dbg_set(21, -1))
    TF.gae_device(*args)  # and the next healthy call is clean again (the counter is cleared per call)


def test_gae_under_graph_capture_replays_correctly(L):
    """A captured launch would replay a frozen per-launch tag, so rlppo_gae switches to its stateless two-launch form while
    the stream is capturing: replays on NEW inputs give the new outputs."""
    rews, dones, trunc, values = synth_gae(64, 256, seed=3)
    n = len(rews)
    vt, adv, ret = (torch.empty(n, device="cuda") for _ in range(3))
    ws = torch.empty(int(L.rlppo_gae_workspace_bytes(n)), dtype=torch.uint8, device="cuda")
    r, d, t, v = dev(rews), dev(dones), dev(trunc), dev(values)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            check(L, L.rlppo_gae(ctypes.c_void_p(side.cuda_stream), P(r), P(d), P(t), P(v), n, 0.99, 0.95, 1.7, P(vt), P(adv), P(ret),
                                 P(ws), ws.numel()))
    for seed in (3, 4, 5):
        rews, dones, trunc, values = synth_gae(64, 256, seed=seed)
        r.copy_(dev(rews)); d.copy_(dev(dones)); t.copy_(dev(trunc)); v.copy_(dev(values))
        g.replay()
        torch.cuda.synchronize()
        _, oadv, oret = ogae.gae(rews, dones, trunc, values, 0.99, 0.95, 1.7, "f64")
        np.testing.assert_allclose(adv.cpu().numpy(), oadv, rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(ret.cpu().numpy(), oret.astype(np.float32), rtol=2e-6, atol=2e-6)


def test_gae_empty(L):
    check(L, L.rlppo_gae(stream(), None, None, None, None, 0, 0.99, 0.95, 1.0, None, None, None, None, 0))


# -------------------------------------------------------------------------------------- network plumbing
class Net:
    def __init__(self, L, params):
        from rlgym_ppo_amd import _native as N
        self.L = L
        self.params = params
        self.dims = [params[0][0].shape[1]] + [w.shape[0] for w, _ in params]
        self.nl = len(params)
        self.dims_c = N.dims_array(self.dims)
        self.flat = dev(nets.flatten(params))
        self.packed = torch.zeros(int(L.rlppo_packed_floats(self.dims_c, self.nl)), device="cuda")
        check(L, L.rlppo_net_pack(stream(), self.dims_c, self.nl, P(self.flat), P(self.packed)))
        self.ld_in = int(L.rlppo_padded_width(self.dims[0]))
        self.ld_out = int(L.rlppo_padded_out(self.dims[-1]))

    def pad(self, obs, f64=False):
        src = dev(obs, torch.float64 if f64 else torch.float32)
        n, d = src.shape
        out = torch.full((n, self.ld_in), float("nan"), device="cuda")
        check(self.L, self.L.rlppo_pad_rows(stream(), P(src), int(f64), n, d, d, P(out), self.ld_in, 0, 0.0, 1.0))
        return out

    def ws(self, n):
        return torch.empty(int(self.L.rlppo_forward_workspace_bytes(self.dims_c, self.nl, n)), dtype=torch.uint8, device="cuda")


def test_g2_value_forward(L, golden):
    g = golden("g2_value_forward")
    net = Net(L, nets.params_from_state(g, "v."))
    for f64 in (False, True):
        rows = net.pad(g["obs"].astype(np.float64) if f64 else g["obs"], f64)
        out = torch.empty(64, net.ld_out, device="cuda")
        w = net.ws(64)
        check(L, L.rlppo_mlp_forward(stream(), net.dims_c, net.nl, P(net.packed), P(rows), net.ld_in, 64, 0, P(out),
                                     net.ld_out, P(w), w.numel(), None))
        assert relerr(out[:, :1], g["values"]) < 1e-5
        assert (out[:, 1:] == 0).all()


@pytest.mark.parametrize("name", ["g1b_discrete_forward_128x2", "g1c_discrete_forward_256x3"])
def test_g1bc_fused_rollout_kernel_against_the_reference(L, golden, name):
    """[r4] The ONE-LAUNCH rollout kernel (csrc/fused_act.hip; what the configs[1] rollout runs) held to vectors the reference itself
    produced -- G1's 32-wide nets never reach it (hidden widths 64 / 128 / 256 only), so until round 4 it was pinned by
    self-comparison and the oracle alone.  Both entry points, and a counter read back from the library says which kernel ran:
    rlppo_discrete_act on padded rows and rlppo_discrete_step on the raw observations; probabilities / log-probabilities within
    1e-5, action indices EXACT (the fixture's smallest selection margin, recorded by the generator, is far above fp32 noise)."""
    g = golden(name)
    assert float(g["margin"].min()) > 1e-4, "fixture holds a near-tie: regenerate with another seed"
    net = Net(L, nets.params_from_state(g, "p."))
    n = g["obs"].shape[0]
    qd = dev(g["q"])
    for entry in ("act", "step"):
        act = torch.empty(n, dtype=torch.int64, device="cuda")
        logp = torch.empty(n, device="cuda")
        fused0, chain0 = int(L.rlppo_dbg_counter(0)), int(L.rlppo_dbg_counter(1))
        if entry == "act":
            rows = net.pad(g["obs"])
            probs = torch.empty(n, 90, device="cuda")
            w = net.ws(n)
            check(L, L.rlppo_discrete_act(stream(), net.dims_c, net.nl, P(net.packed), P(rows), net.ld_in, n, P(qd), P(act), P(logp),
                                          P(probs), P(w), w.numel(), None))
            assert relerr(probs, g["probs"]) < 1e-5
        else:
            raw = dev(g["obs"])
            w = torch.empty(int(L.rlppo_discrete_step_workspace_bytes(net.dims_c, net.nl, n)), dtype=torch.uint8, device="cuda")
            check(L, L.rlppo_discrete_step(stream(), net.dims_c, net.nl, P(net.packed), P(raw), 0, raw.shape[1], n, 0, 0.0, 1.0, None, None,
                                           P(qd), P(act), None, P(logp), None, 0, P(w), w.numel(), None))
        assert int(L.rlppo_dbg_counter(0)) == fused0 + 1 and int(L.rlppo_dbg_counter(1)) == chain0, "the fused kernel did not run"
        assert np.array_equal(act.cpu().numpy(), g["actions"]), entry
        assert np.abs(logp.cpu().numpy() - g["logp"]).max() < 1e-5, entry
    # and the layer chain on the same vectors (rlppo_dbg_set(27, 0)): the two forms are bit-identical to each other
    check(L, L.rlppo_dbg_set(27, 0))
    try:
        act2, logp2 = torch.empty(n, dtype=torch.int64, device="cuda"), torch.empty(n, device="cuda")
        rows, w = net.pad(g["obs"]), net.ws(n)
        chain0 = int(L.rlppo_dbg_counter(1))
        check(L, L.rlppo_discrete_act(stream(), net.dims_c, net.nl, P(net.packed), P(rows), net.ld_in, n, P(qd), P(act2), P(logp2), None,
                                      P(w), w.numel(), None))
        assert int(L.rlppo_dbg_counter(1)) == chain0 + 1
    finally:
        check(L, L.rlppo_dbg_set(27, 1))
    assert torch.equal(act2, act) and torch.equal(logp2, logp)


def test_g1_discrete_act(L, golden):
    g = golden("g1_discrete_forward")
    net = Net(L, nets.params_from_state(g, "p."))
    rows = net.pad(g["obs"])
    act = torch.empty(64, dtype=torch.int64, device="cuda")
    logp = torch.empty(64, device="cuda")
    probs = torch.empty(64, 90, device="cuda")
    w = net.ws(64)
    qd, pd = dev(g["q"]), dev(g["probs"])
    check(L, L.rlppo_discrete_act(stream(), net.dims_c, net.nl, P(net.packed), P(rows), net.ld_in, 64, P(qd),
                                  P(act), P(logp), P(probs), P(w), w.numel(), None))
    assert relerr(probs, g["probs"]) < 1e-5
    assert np.abs(logp.cpu().numpy() - g["logp"]).max() < 1e-5
    assert np.array_equal(act.cpu().numpy(), g["actions"])
    # the selection step itself is exact: fed the reference's own probabilities it must return the reference's indices
    act2 = torch.empty(64, dtype=torch.int64, device="cuda")
    lp2 = torch.empty(64, device="cuda")
    check(L, L.rlppo_categorical_select(stream(), P(pd), 90, 64, 90, P(qd), P(act2), P(lp2)))
    assert np.array_equal(act2.cpu().numpy(), g["actions"])
    np.testing.assert_allclose(lp2.cpu().numpy(), g["logp"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("d,hidden,A", [(107, (256, 256, 256), 90), (107, (64, 64), 90), (20, (128, 128, 128, 128), 7), (231, (256, 256), 128),
                                        (13, (64,), 5)])
def test_fused_rollout_step_is_bit_identical_to_the_layer_chain(L, d, hidden, A):
    """[r3] rlppo_discrete_act as ONE launch (csrc/fused_act.hip, SURVEY K1: MLP -> softmax -> clamp -> argmax(p/q) -> log p with the
    activations in LDS and every wave streaming its own weight rows) against the layer-by-layer chain it replaces
    (rlppo_dbg_set(27, 0)): clamped probabilities, actions and log-probabilities must agree BIT for bit, for one row, exactly one
    row tile, ragged counts and the 4096-row rollout shape; several launches in a row (the weight ring's tail waits)."""
    torch.manual_seed(d + A)
    net = Net(L, nets.init_mlp(d, hidden, A))
    rs = np.random.RandomState(d)
    for n in (1, 16, 17, 250, 4096, 5000, 8192):
        rows = net.pad(np.clip(rs.randn(n, d) * 2, -5, 5).astype(np.float32))
        q = dev(rs.exponential(size=(n, A)).astype(np.float32))
        outs = []
        for fused in (1, 0, 1):
            act = torch.full((n,), -7, dtype=torch.int64, device="cuda")
            logp = torch.full((n,), float("nan"), device="cuda")
            probs = torch.full((n, A), float("nan"), device="cuda")
            w = net.ws(n)
            check(L, L.rlppo_dbg_set(27, fused))
            try:
                check(L, L.rlppo_discrete_act(stream(), net.dims_c, net.nl, P(net.packed), P(rows), net.ld_in, n, P(q), P(act), P(logp),
                                              P(probs), P(w), w.numel(), None))
            finally:
                check(L, L.rlppo_dbg_set(27, 1))
            outs.append((act.cpu(), logp.cpu(), probs.cpu()))
        for a, lp, pr in outs[1:]:
            assert torch.equal(outs[0][0], a) and torch.equal(outs[0][1], lp) and torch.equal(outs[0][2], pr), (n, d, hidden, A)
        assert (outs[0][0] >= 0).all() and (outs[0][0] < A).all() and not torch.isnan(outs[0][2]).any()
    # and against the oracle at the rollout shape (the chain's own test bound: near-ties only)
    n = 512
    obs = np.clip(rs.randn(n, d), -5, 5).astype(np.float32)
    qh = nets.draw_exp_noise(n, A)
    act = torch.empty(n, dtype=torch.int64, device="cuda")
    logp = torch.empty(n, device="cuda")
    w = net.ws(n)
    check(L, L.rlppo_discrete_act(stream(), net.dims_c, net.nl, P(net.packed), P(net.pad(obs)), net.ld_in, n, P(dev(qh)), P(act), P(logp),
                                  None, P(w), w.numel(), None))
    oact, ologp = nets.discrete_sample(nets.discrete_probs(net.params, obs), qh)
    same = act.cpu() == oact
    assert (~same).sum().item() <= 2 and (logp.cpu() - ologp)[same].abs().max().item() < 1e-5


@pytest.mark.parametrize("hidden", [(256, 256, 256), (96, 96)], ids=["fused256", "chain_only"])
def test_act_options_completion_words_and_per_call_precision(L, hidden):
    """[r5] rlppo_act_opts: (a) done_words -- the call's last kernel stores one word per 16 rows into pinned host memory behind the
    results (the one-launch kernel workgroup by workgroup, the layer chain through one more tiny launch); a host that polls them
    with rlppo_host_wait_words reads the SAME actions / log-probabilities a stream synchronisation would have delivered, without
    synchronising; a wait for a value nobody stores times out (return 1), never hangs.  (b) precision -- the inference precision
    is an argument of the call: bf16 operands through the option == bf16 operands through the process-wide switch, bit for bit,
    and a call without the option is untouched by another call's option."""
    from rlgym_ppo_amd import _native as N
    d, A, n = 107, 90, 200
    torch.manual_seed(5 + hidden[0])
    net = Net(L, nets.init_mlp(d, hidden, A))
    rs = np.random.RandomState(hidden[0])
    obs = torch.from_numpy(np.clip(rs.randn(n, d), -5, 5).astype(np.float32)).pin_memory()     # host inputs, as ActGraph hands them over
    q = torch.from_numpy(rs.exponential(size=(n, A)).astype(np.float32)).pin_memory()
    ws = torch.empty(int(L.rlppo_discrete_step_workspace_bytes(net.dims_c, net.nl, n)), dtype=torch.uint8, device="cuda")
    n_words = int(L.rlppo_act_done_words(n))
    assert n_words == 13 and L.rlppo_act_done_words(0) == 0 and L.rlppo_act_done_words(16) == 1 and L.rlppo_act_done_words(17) == 2

    def step(opts, act, logp):
        check(L, L.rlppo_discrete_step(stream(), net.dims_c, net.nl, P(net.packed), P(obs), 0, d, n, 0, 0.0, 1.0, None, None, P(q), P(act), None,
                                       P(logp), None, 0, P(ws), ws.numel(), opts))

    ref_a, ref_l = torch.empty(n, dtype=torch.int64).pin_memory(), torch.empty(n).pin_memory()
    step(None, ref_a, ref_l)
    torch.cuda.synchronize()
    done = torch.zeros(n_words, dtype=torch.int32).pin_memory()
    for value in (7, 8, 0x7FFFFFFF):
        act, logp = torch.full((n,), -1, dtype=torch.int64).pin_memory(), torch.full((n,), float("nan")).pin_memory()
        opts = N.ActOpts(N.PRECISION_DEFAULT, value, done.data_ptr())
        step(ctypes.byref(opts), act, logp)
        assert L.rlppo_host_wait_words(P(done), n_words, value, 5_000_000) == 0      # no stream synchronisation before the reads below
        assert torch.equal(act, ref_a) and torch.equal(logp, ref_l), value
        assert (done.numpy() == value).all()
    torch.cuda.synchronize()
    assert L.rlppo_host_wait_words(P(done), n_words, 12345, 20_000) == 1             # nobody stores that value: a bounded wait
    assert L.rlppo_host_wait_words(P(done), 0, 1, 0) == 0
    # (b) per-call precision
    bad = N.ActOpts(7, 0, None)
    a0, l0 = torch.empty(n, dtype=torch.int64, device="cuda"), torch.empty(n, device="cuda")
    assert L.rlppo_discrete_step(stream(), net.dims_c, net.nl, P(net.packed), P(obs), 0, d, n, 0, 0.0, 1.0, None, None, P(q), P(a0), None, P(l0), None,
                                 0, P(ws), ws.numel(), ctypes.byref(bad)) == 1001
    rows = net.pad(obs.numpy())
    w = net.ws(n)
    outs = {}
    for key, glob, opt in (("fp32", 0, None), ("bf16_switch", 1, None), ("bf16_option", 0, N.PRECISION_BF16), ("fp32_option_under_bf16_switch", 1, N.PRECISION_FP32)):
        check(L, L.rlppo_set_inference_precision(glob))
        try:
            out = torch.empty(n, net.ld_out, device="cuda")
            o = N.ActOpts(opt, 0, None) if opt is not None else None
            check(L, L.rlppo_mlp_forward(stream(), net.dims_c, net.nl, P(net.packed), P(rows), net.ld_in, n, 0, P(out), net.ld_out, P(w), w.numel(),
                                         ctypes.byref(o) if o is not None else None))
            outs[key] = out.cpu()
        finally:
            check(L, L.rlppo_set_inference_precision(0))
    assert torch.equal(outs["bf16_switch"], outs["bf16_option"]) and torch.equal(outs["fp32"], outs["fp32_option_under_bf16_switch"])
    assert not torch.equal(outs["fp32"], outs["bf16_option"]) and relerr(outs["bf16_option"], outs["fp32"]) < 3e-2


@pytest.mark.parametrize("n", [1, 8, 17, 80, 300, 1024])
def test_gemm_nt_skinny(L, n):
    """[r5] The forward layers of a small call (up to 1024 rows) run as one wave per 16 x 16 output block with both operands straight
    from L2 (gemm_nt_skinny_kernel) instead of 128-row tiles that are mostly padding at 8-80 rows: outputs BIT-identical to the
    large kernel's (rlppo_dbg_set(40, 0)) for every layer shape of the three heads -- first layers padded to 112 / 240, hidden 256 /
    512, heads of 90 (padded 96), 16 + tanh, 32 -- and against the oracle's float64 forward within fp32 accumulation error."""
    rs = np.random.RandomState(n)
    for d, hidden, out, tanh in ((107, (256, 256, 256), 90, 0), (231, (512, 512, 512, 512), 16, 1), (107, (64, 128), 24, 0), (40, (512,), 90, 0)):
        torch.manual_seed(d + n)
        net = Net(L, nets.init_mlp(d, hidden, out))
        obs = np.clip(rs.randn(n, d), -5, 5).astype(np.float32)
        rows = net.pad(obs)
        outs = []
        for skinny in (1, 0):
            check(L, L.rlppo_dbg_set(40, skinny))
            try:
                o = torch.full((n, net.ld_out), float("nan"), device="cuda")
                w = net.ws(n)
                check(L, L.rlppo_mlp_forward(stream(), net.dims_c, net.nl, P(net.packed), P(rows), net.ld_in, n, tanh, P(o), net.ld_out, P(w), w.numel(), None))
                outs.append(o.cpu())
            finally:
                check(L, L.rlppo_dbg_set(40, 1))
        assert torch.equal(outs[0], outs[1]), (n, d, hidden)
        h = torch.from_numpy(obs).double()
        for i, (w_, b_) in enumerate(net.params):       # the reference's op sequence (value_estimator.py:30-36) in float64
            h = torch.nn.functional.linear(h, w_.double(), b_.double())
            h = torch.relu(h) if i < len(net.params) - 1 else (torch.tanh(h) if tanh else h)
        assert relerr(outs[0][:, :out].double(), h) < 2e-5


@pytest.mark.parametrize("where", ["host_window", "pinned"])
@pytest.mark.parametrize("hidden", [(256, 256, 256), (128, 128)], ids=["fused256", "fused128"])
def test_discrete_step_takes_its_noise_while_it_runs(L, hidden, where):
    """[r5] rlppo_act_opts.noise_ctl: the one-launch step is launched BEFORE its noise exists; the host then fills the noise matrix
    and stores the call's sequence into control word 2 (rlppo_host_push: bytes, fence, word, fence); the kernel -- which looks for
    that word when its head layer starts and waits for it after the last layer if it was not there -- samples with exactly those
    numbers: actions / log-probabilities bit-identical to the call that had its noise staged beforehand, for a host that completes
    at once and for one that takes 2 ms; rows past the live count are untouched; a host that does not complete gets completion
    words with the failure bit after 20 ms (rlppo_host_wait_words returns 2), not a hung GPU, and the same call made again after
    the completion delivers; entry points / networks without the one-launch kernel refuse the option.  Observations, noise and
    control words live in a host window (device memory the host writes through the PCIe aperture: what ActGraph uses) or in
    pinned host memory."""
    import time
    from rlgym_ppo_amd import _native as N
    d, A, n, live = 107, 90, 96, 83
    torch.manual_seed(11 + hidden[0])
    net = Net(L, nets.init_mlp(d, hidden, A))
    rs = np.random.RandomState(hidden[0] + 1)
    obs_h = np.clip(rs.randn(n, d), -5, 5).astype(np.float32)
    ws = torch.empty(int(L.rlppo_discrete_step_workspace_bytes(net.dims_c, net.nl, n)), dtype=torch.uint8, device="cuda")
    n_words = int(L.rlppo_act_done_words(n))
    done = torch.zeros(n_words, dtype=torch.int32).pin_memory()
    win = ctypes.c_void_p()
    if where == "host_window":
        check(L, L.rlppo_host_window_alloc(256 + 4 * n * d + 4 * n * A, ctypes.byref(win)))
        ctl_p, obs_p, q_p = win.value, win.value + 256, win.value + 256 + 4 * n * d
    else:
        keep = (torch.zeros(32, dtype=torch.int32).pin_memory(), torch.zeros(n, d).pin_memory(), torch.ones(n, A).pin_memory())
        ctl_p, obs_p, q_p = (t.data_ptr() for t in keep)
    V = ctypes.c_void_p
    check(L, L.rlppo_host_push(V(obs_p), V(obs_h.ctypes.data), obs_h.nbytes, None, 0))
    obs_pin = torch.from_numpy(obs_h).pin_memory()
    hdr = np.zeros(2, dtype=np.uint32)

    def step(opts, obs_ptr, noise_ptr, act, logp):
        return L.rlppo_discrete_step(stream(), net.dims_c, net.nl, P(net.packed), V(obs_ptr), 0, d, n, 0, 0.0, 1.0, None, None, V(noise_ptr), P(act), None,
                                     P(logp), None, 0, P(ws), ws.numel(), opts)

    try:
        opts = N.ActOpts(N.PRECISION_DEFAULT, 1, done.data_ptr(), ctl_p)
        assert L.rlppo_discrete_step_one_launch(net.dims_c, net.nl, n, ctypes.byref(opts)) == 1
        for seq, delay in ((1, 0.0), (2, 0.002), (0x7FFFFFFF, 0.0), (3, 0.0)):
            q = torch.from_numpy(rs.exponential(size=(n, A)).astype(np.float32)).pin_memory()
            ref_a, ref_l = torch.empty(n, dtype=torch.int64).pin_memory(), torch.empty(n).pin_memory()
            check(L, step(None, obs_pin.data_ptr(), q.data_ptr(), ref_a, ref_l))
            torch.cuda.synchronize()
            act, logp = torch.full((n,), -1, dtype=torch.int64).pin_memory(), torch.full((n,), float("nan")).pin_memory()
            done.zero_()
            hdr[:] = (seq, live)
            check(L, L.rlppo_host_push(V(ctl_p), V(hdr.ctypes.data), 8, None, 0))
            check(L, L.rlppo_host_window_flush(V(ctl_p)))                   # (a no-op for memory that is not a host window)
            opts.done_value = seq
            check(L, step(ctypes.byref(opts), obs_p, q_p, act, logp))       # (the noise matrix still holds the previous round's numbers)
            if delay:
                time.sleep(delay)
                assert L.rlppo_host_wait_words(P(done), n_words, seq, 0) == 1   # still waiting for its noise
            check(L, L.rlppo_host_push(V(q_p), V(q.data_ptr()), 4 * live * A, V(ctl_p + 8), seq))
            assert L.rlppo_host_wait_words(P(done), n_words, seq, 5_000_000) == 0
            assert torch.equal(act[:live], ref_a[:live]) and torch.equal(logp[:live], ref_l[:live]), (seq, delay)
            assert (act[live:] == -1).all() and torch.isnan(logp[live:]).all()
        torch.cuda.synchronize()
        # a host that does not complete: the kernel gives up after 20 ms and says so; the call made again after the completion delivers
        done.zero_()
        hdr[:] = (77, live)
        check(L, L.rlppo_host_push(V(ctl_p), V(hdr.ctypes.data), 8, None, 0))
        opts.done_value = 5
        act, logp = torch.full((n,), -1, dtype=torch.int64).pin_memory(), torch.full((n,), float("nan")).pin_memory()
        t0 = time.time()
        check(L, step(ctypes.byref(opts), obs_p, q_p, act, logp))
        assert L.rlppo_host_wait_words(P(done), n_words, 5, 10_000_000) == 2
        torch.cuda.synchronize()
        assert 0.015 < time.time() - t0 < 1.0
        assert (done.numpy().astype(np.uint32) == np.uint32(5 | 0x80000000)).all()
        check(L, L.rlppo_host_push(None, None, 0, V(ctl_p + 8), 77))
        done.zero_()
        check(L, step(ctypes.byref(opts), obs_p, q_p, act, logp))
        assert L.rlppo_host_wait_words(P(done), n_words, 5, 5_000_000) == 0
        assert torch.equal(act[:live], ref_a[:live]) and torch.equal(logp[:live], ref_l[:live])
        # refused where no one-launch kernel runs and without completion words
        a0, l0 = torch.empty(n, dtype=torch.int64, device="cuda"), torch.empty(n, device="cuda")
        rows = net.pad(obs_h)
        w = net.ws(n)
        assert L.rlppo_discrete_act(stream(), net.dims_c, net.nl, P(net.packed), P(rows), net.ld_in, n, V(q_p), P(a0), P(l0), None, P(w), w.numel(),
                                    ctypes.byref(opts)) == 1001
        no_words = N.ActOpts(N.PRECISION_DEFAULT, 1, None, ctl_p)
        assert step(ctypes.byref(no_words), obs_p, q_p, act, logp) == 1001
        check(L, L.rlppo_dbg_set(27, 0))   # the layer chain instead of the one-launch kernel
        try:
            assert L.rlppo_discrete_step_one_launch(net.dims_c, net.nl, n, ctypes.byref(opts)) == 0
            assert step(ctypes.byref(opts), obs_p, q_p, act, logp) == 1001
        finally:
            check(L, L.rlppo_dbg_set(27, 1))
        torch.cuda.synchronize()
    finally:
        torch.cuda.synchronize()
        if win.value:
            check(L, L.rlppo_host_window_free(win))


@pytest.mark.parametrize("hidden", [(256, 256, 256), (64, 64), (96, 96)], ids=["fused256", "fused64", "chain_only"])
def test_discrete_step_raw_observations_all_modes(L, hidden):
    """[r3] rlppo_discrete_step -- raw observations (fp32 / fp64, ragged row stride) -> standardise (none / the reference's scalars of
    feature 0 / per-feature vectors) + pad -> policy -> actions, the actions as floats, log-probs and the padded rows -- against the
    launch-by-launch form it replaces (rlppo_pad_rows[_per_feature] + rlppo_discrete_act on the layer chain, rlppo_dbg_set(27, 0)):
    every output BIT-identical, whether the network has the fused kernel's form (256- and 64-wide) or not (96-wide: the entry point
    then runs the chain itself)."""
    d, A, n = 107, 90, 777
    torch.manual_seed(len(hidden) + hidden[0])
    net = Net(L, nets.init_mlp(d, hidden, A))
    rs = np.random.RandomState(hidden[0])
    raw = (rs.randn(n, d + 5) * 3 + 0.7)          # ld_obs = d + 5: the observations are a column slice of a wider array
    q = dev(rs.exponential(size=(n, A)).astype(np.float32))
    mean_v, std_v = dev(rs.randn(d).astype(np.float32)), dev((0.5 + rs.rand(d)).astype(np.float32))
    ws = torch.empty(int(L.rlppo_discrete_step_workspace_bytes(net.dims_c, net.nl, n)), dtype=torch.uint8, device="cuda")
    for f64 in (False, True):
        obs = dev(raw, torch.float64 if f64 else torch.float32)
        for mode in (0, 1, 2):
            outs = []
            for fused in (1, 0):
                act = torch.full((n,), -1, dtype=torch.int64, device="cuda")
                actf = torch.full((n,), float("nan"), device="cuda")
                logp = torch.full((n,), float("nan"), device="cuda")
                rows = torch.full((n, net.ld_in + 16), float("nan"), device="cuda")   # ld_rows_out wider than the padded width
                check(L, L.rlppo_dbg_set(27, fused))
                try:
                    check(L, L.rlppo_discrete_step(stream(), net.dims_c, net.nl, P(net.packed), P(obs), int(f64), d + 5, n, mode, 0.3, 1.7,
                                                    P(mean_v) if mode == 2 else None, P(std_v) if mode == 2 else None, P(q), P(act), P(actf),
                                                    P(logp), P(rows), rows.shape[1], P(ws), ws.numel(), None))
                finally:
                    check(L, L.rlppo_dbg_set(27, 1))
                outs.append((act.cpu(), actf.cpu(), logp.cpu(), rows[:, :net.ld_in].cpu()))
            for a, b in zip(outs[0], outs[1]):
                assert torch.equal(a, b), (hidden, f64, mode)
            act, actf, logp, rows = outs[0]
            assert torch.equal(actf, act.float()) and (act >= 0).all() and (act < A).all()
            # the padded rows: the reference's expression (batched_agent_manager.py:313-315) in fp32, zero padding
            x = obs[:, :d].float().cpu()
            want = x if mode == 0 else ((x - 0.3) / 1.7).clamp(-5, 5) if mode == 1 else ((x - mean_v.cpu()) / std_v.cpu()).clamp(-5, 5)
            assert torch.allclose(rows[:, :d], want, rtol=2e-6, atol=2e-6) and (rows[:, d:] == 0).all()
            if mode == 0:
                assert torch.equal(rows[:, :d], want)
    assert L.rlppo_discrete_step(stream(), net.dims_c, net.nl, P(net.packed), P(obs), 0, d - 1, n, 0, 0.0, 1.0, None, None, P(q), P(act), None,
                                 P(logp), None, 0, P(ws), ws.numel(), None) != 0   # ld_obs < d is refused


@pytest.mark.parametrize("A", [90, 300, 1000])
def test_discrete_probs_and_deterministic_choice(L, golden, A):
    """rlppo_discrete_probs = DiscreteFF.get_output (softmax) and the deterministic branch of get_action: clamp(1e-11, 1), then
    numpy's argmax over the FLATTENED [n, A] array (discrete_policy.py:52-57, quirk Q11) -- first occurrence of the maximum."""
    g = golden("g1_discrete_forward")
    if A == 90:  # the reference's own numbers
        net = Net(L, nets.params_from_state(g, "p."))
        rows = net.pad(g["obs"])
        probs, best, w = torch.full((64, 96), float("nan"), device="cuda"), torch.full((1,), -7, dtype=torch.int64, device="cuda"), net.ws(64)
        check(L, L.rlppo_discrete_probs(stream(), net.dims_c, net.nl, P(net.packed), P(rows), net.ld_in, 64, 1, P(probs), 96, P(best),
                                        P(w), w.numel(), None))
        assert relerr(probs[:, :90], g["probs"]) < 1e-5 and torch.isnan(probs[:, 90:]).all()  # ld_probs respected
        assert int(best) == int(g["det_action"]) == int(probs[:, :90].cpu().numpy().argmax())
    rs = np.random.RandomState(A)
    n, d = 777, 40
    params = [(torch.as_tensor(rs.randn(A, d).astype(np.float32)), torch.as_tensor(rs.randn(A).astype(np.float32) * 3))]
    params[0][1][5] = 60.0  # softmax underflows below 1e-11 elsewhere in those rows -> the clamp is visible
    net = Net(L, params)
    obs = rs.randn(n, d).astype(np.float32)
    rows, w = net.pad(obs), net.ws(n)
    soft, clamped = torch.empty(n, A, device="cuda"), torch.empty(n, A, device="cuda")
    best = torch.empty(1, dtype=torch.int64, device="cuda")
    check(L, L.rlppo_discrete_probs(stream(), net.dims_c, net.nl, P(net.packed), P(rows), net.ld_in, n, 0, P(soft), A, None, P(w), w.numel(), None))
    check(L, L.rlppo_discrete_probs(stream(), net.dims_c, net.nl, P(net.packed), P(rows), net.ld_in, n, 1, P(clamped), A, P(best), P(w), w.numel(), None))
    ref = torch.softmax(nets.mlp(params, torch.as_tensor(obs)).double(), -1)
    assert relerr(soft, ref) < 2e-6 and float(soft.min()) < 1e-11
    assert torch.equal(clamped, soft.clamp(min=1e-11, max=1))
    assert int(best) == int(clamped.cpu().numpy().argmax())  # exact given the library's own probabilities
    # ties: a bias-only head gives every row the same probabilities with the maximum at two columns -> the first column of row 0
    bias = torch.zeros(A)
    bias[[A - 3, 17]] = 4.0
    tie = Net(L, [(torch.zeros(A, d), bias)])
    only = torch.empty(1, dtype=torch.int64, device="cuda")
    check(L, L.rlppo_discrete_probs(stream(), tie.dims_c, tie.nl, P(tie.packed), P(rows), tie.ld_in, n, 1, None, 0, P(only), P(w), w.numel(), None))
    assert int(only) == 17
    assert L.rlppo_discrete_probs(stream(), tie.dims_c, tie.nl, P(tie.packed), P(rows), tie.ld_in, n, 1, None, 0, None, P(w), w.numel(), None) != 0


def test_categorical_select_exact_at_scale(L):
    # 4096 x 90 (the rollout shape of BASELINE configs[1]): identical probs + identical noise => identical indices
    torch.manual_seed(3)
    probs = torch.softmax(torch.randn(4096, 90) * 3, -1).clamp(1e-11, 1)
    q = torch.empty(4096, 90).exponential_(1)
    ref = torch.argmax(probs / q, -1)
    act = torch.empty(4096, dtype=torch.int64, device="cuda")
    lp = torch.empty(4096, device="cuda")
    pd, qd = dev(probs), dev(q)
    check(L, L.rlppo_categorical_select(stream(), P(pd), 90, 4096, 90, P(qd), P(act), P(lp)))
    assert torch.equal(act.cpu(), ref)
    # ties: first index wins
    probs = torch.full((4, 90), 1.0 / 90)
    q = torch.ones(4, 90)
    pd, qd = dev(probs), dev(q)
    check(L, L.rlppo_categorical_select(stream(), P(pd), 90, 4, 90, P(qd), P(act), P(lp)))
    assert (act[:4].cpu() == 0).all()


def test_rollout_shape_discrete_act_vs_oracle(L):
    # configs[1] shape: 4096 agents x obs 107, 256x3 MLP, 90 actions
    torch.manual_seed(123)
    pol = nets.init_mlp(107, (256, 256, 256), 90)
    rs = np.random.RandomState(0)
    obs = np.clip(rs.randn(4096, 107), -5, 5).astype(np.float32)
    q = nets.draw_exp_noise(4096, 90)
    oprobs = nets.discrete_probs(pol, obs)
    oact, ologp = nets.discrete_sample(oprobs, q)
    net = Net(L, pol)
    rows = net.pad(obs)
    act = torch.empty(4096, dtype=torch.int64, device="cuda")
    logp = torch.empty(4096, device="cuda")
    probs = torch.empty(4096, 90, device="cuda")
    w = net.ws(4096)
    qd = dev(q)
    check(L, L.rlppo_discrete_act(stream(), net.dims_c, net.nl, P(net.packed), P(rows), net.ld_in, 4096, P(qd), P(act),
                                  P(logp), P(probs), P(w), w.numel(), None))
    assert relerr(probs, oprobs) < 1e-5
    # with its own (ulp-different) probs the indices agree except on near-ties: margin stated = 1e-4 relative in p/q
    a = act.cpu()
    diff = (a != oact).nonzero().flatten()
    score = (oprobs / q)
    for i in diff.tolist():
        s = score[i]
        assert abs(s[a[i]] - s[oact[i]]) <= 1e-4 * s[oact[i]], "index mismatch that is not a near-tie"
    assert len(diff) <= 2
    same = a == oact
    assert np.abs(logp.cpu().numpy()[same] - ologp.numpy()[same]).max() < 1e-5


def test_g9_gaussian_and_multidiscrete_act(L, golden):
    g = golden("g9_continuous")
    net = Net(L, nets.params_from_state(g, "p."))
    rows = net.pad(g["obs"])
    n = 80
    act = torch.empty(n, 8, device="cuda")
    logp = torch.empty(n, device="cuda")
    w = net.ws(n)
    m, b = nets.var_map(0.1, 1.0)
    epsd = dev(g["eps"])
    check(L, L.rlppo_gaussian_act(stream(), net.dims_c, net.nl, P(net.packed), P(rows), net.ld_in, n, P(epsd), m, b,
                                  P(act), P(logp), P(w), w.numel(), None))
    np.testing.assert_allclose(act.cpu().numpy(), g["act"], rtol=1e-5, atol=2e-6)
    clamped = np.abs(g["act"]) == 1.0
    assert np.array_equal(np.abs(act.cpu().numpy()) == 1.0, clamped)
    # log-probabilities: float64 truth with the bound the 4-term formula's conditioning implies (tests/fp64_gate.py), and the
    # reference's own float32 numbers within the sum of the two implementations' bounds
    y = torch.empty(n, net.ld_out, device="cuda")
    check(L, L.rlppo_mlp_forward(stream(), net.dims_c, net.nl, P(net.packed), P(rows), net.ld_in, n, 1, P(y), net.ld_out, P(w), w.numel(), None))
    res = fp64_gate.gauss_logp_check(net.params, g["obs"], g["eps"], y, act, logp, label="G9")
    assert np.abs(logp.cpu().numpy() - g["logp"]).max() <= res["hip"][2] * (res["hip"][1] + res["cpu"][1]) + 1e-7

    g = golden("g9_multidiscrete")
    net = Net(L, nets.params_from_state(g, "p."))
    rows = net.pad(g["obs"])
    n = 72
    act = torch.empty(n, 8, dtype=torch.int64, device="cuda")
    logp = torch.empty(n, device="cuda")
    w = net.ws(n)
    qd = dev(g["q"])
    check(L, L.rlppo_multidiscrete_act(stream(), net.dims_c, net.nl, P(net.packed), P(rows), net.ld_in, n, P(qd),
                                       P(act), P(logp), P(w), w.numel(), None))
    assert np.array_equal(act.cpu().numpy(), g["act"])
    np.testing.assert_allclose(logp.cpu().numpy(), g["logp"], rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------------------------------- PPO minibatch
def run_minibatch(L, head, pol, val, obs_all, acts_all, old_all, tgt_all, adv_all, idx, clip, ent, mb_ratio, var=(0.1, 1.0),
                  precision="fp32"):
    from rlgym_ppo_amd import _native as N
    if precision == "bf16":  # the bf16-operand forward (rlppo_set_update_precision): same call with the rounded weight images
        check(L, L.rlppo_set_update_precision(1))
        try:
            return run_minibatch(L, head, pol, val, obs_all, acts_all, old_all, tgt_all, adv_all, idx, clip, ent, mb_ratio, var, "bf16*")
        finally:
            check(L, L.rlppo_set_update_precision(0))
    if precision == "x3":  # [r4] the split-bf16 hidden products (rlppo_set_update_precision(2)): same call with the three-plane images
        check(L, L.rlppo_set_update_precision(2))
        try:
            return run_minibatch(L, head, pol, val, obs_all, acts_all, old_all, tgt_all, adv_all, idx, clip, ent, mb_ratio, var, "x3*")
        finally:
            check(L, L.rlppo_set_update_precision(0))
    P_, V_ = Net(L, pol), Net(L, val)
    states = P_.pad(obs_all)
    acts = dev(np.asarray(acts_all, np.float32).reshape(len(obs_all), -1))
    a = N.MinibatchArgs()
    a.head = {"discrete": 0, "multidiscrete": 1, "gaussian": 2}[head]
    a.pol_layers, a.val_layers, a.act_dim = P_.nl, V_.nl, acts.shape[1]
    a.pol_dims = ctypes.cast(P_.dims_c, ctypes.POINTER(ctypes.c_int32))
    a.val_dims = ctypes.cast(V_.dims_c, ctypes.POINTER(ctypes.c_int32))
    gp = torch.zeros_like(P_.flat)
    gv = torch.zeros_like(V_.flat)
    old, tgt, adv = dev(old_all), dev(tgt_all), dev(adv_all)
    idxd = dev(idx, torch.int64)
    stats = torch.zeros(8, dtype=torch.float64, device="cuda")
    mb = len(idx)
    ws = torch.empty(int(L.rlppo_minibatch_workspace_bytes(P_.dims_c, P_.nl, V_.dims_c, V_.nl, mb)), dtype=torch.uint8, device="cuda")
    a.pol_packed, a.val_packed, a.pol_grad, a.val_grad = P_.packed.data_ptr(), V_.packed.data_ptr(), gp.data_ptr(), gv.data_ptr()
    if precision == "bf16*":
        imgs = []
        for net in (P_, V_):
            pr = torch.zeros_like(net.packed)
            wb = torch.zeros(int(L.rlppo_wb16_elems(net.dims_c, net.nl)), dtype=torch.bfloat16, device="cuda")
            check(L, L.rlppo_net_pack_bf16(stream(), net.dims_c, net.nl, P(net.flat), P(pr), P(wb)))
            imgs += [pr, wb]
        a.pol_packed_r, a.pol_wb16, a.val_packed_r, a.val_wb16 = (t.data_ptr() for t in imgs)
    if precision == "x3*":
        imgs = []
        for net in (P_, V_):
            pl = torch.zeros(max(int(L.rlppo_x3_elems(net.dims_c, net.nl)), 8), dtype=torch.bfloat16, device="cuda")
            check(L, L.rlppo_net_pack_x3(stream(), net.dims_c, net.nl, P(net.packed), P(pl)))
            imgs.append(pl)
        a.pol_wb16, a.val_wb16 = (t.data_ptr() for t in imgs)
    a.states, a.ld_states, a.n_rows, a.actions = states.data_ptr(), states.shape[1], states.shape[0], acts.data_ptr()
    a.old_logp, a.targets, a.advantages, a.idx, a.mb = old.data_ptr(), tgt.data_ptr(), adv.data_ptr(), idxd.data_ptr(), mb
    a.clip_range, a.ent_coef, a.mb_ratio = clip, ent, mb_ratio
    a.var_m, a.var_b = nets.var_map(*var)
    a.stats, a.workspace, a.ws_bytes = stats.data_ptr(), ws.data_ptr(), ws.numel()
    check(L, L.rlppo_ppo_minibatch(stream(), ctypes.byref(a)))
    torch.cuda.synchronize()
    return nets.unflatten(gp.cpu(), pol), nets.unflatten(gv.cpu(), val), stats.cpu().numpy()


def test_float64_truth_is_guarded_on_the_gpu_box_too(golden):
    """Every GPU gate leans on oracle/ppo.py::minibatch_analytic (float64, hand-derived gradients = the formulas of csrc/heads.hip).
    What breaks the common mode between it and the kernels is its check against the AUTOGRAD form of the same minibatch -- a CPU
    test (tests/test_oracle_nets_ppo.py); it runs here as well, on the GPU box's host (another CPU, other ATen kernels), so that
    the truth the `-m gpu` set measures against is verified in the same run."""
    import test_oracle_nets_ppo as cpu
    cpu.test_g4_discrete_loss_and_grads(golden)
    cpu.test_g9_gaussian_head(golden)
    cpu.test_g9_multidiscrete_head(golden)


def test_g4_discrete_minibatch_against_reference(L, golden):
    """Reference fixture G4 as it is -- 32 rows engineered onto the clip edges, 32 exact torch.min ties at ratio == 1, clamped
    probabilities, clipped-high / clipped-low rows -- against float64 truth, every row in (tests/fp64_gate.py); and the parts of
    the reference's own float32 output that no knife-edge row touches, directly."""
    g = golden("g4_discrete_loss")
    pol, val = nets.params_from_state(g, "p."), nets.params_from_state(g, "v.")
    idx = np.arange(96)
    got = run_minibatch(L, "discrete", pol, val, g["obs"], g["acts"], g["old_logp"], g["targets"], g["adv"], idx, 0.2, 0.005, 0.25)
    out = fp64_gate.gate(L, "discrete", pol, val, g["obs"], g["acts"], g["old_logp"], g["adv"], g["targets"], 0.2, 0.005, 0.25, got,
                         label="G4 (reference fixture, 32 clip-edge rows)")
    assert out["edge_rows"] == 32
    gp2, gv2, stats2 = got
    assert abs(stats2[0] - float(g["out.entropy"])) < 1e-5 * abs(float(g["out.entropy"]))
    assert abs(stats2[2] - float(g["out.value_loss"])) < 1e-5 * abs(float(g["out.value_loss"]))
    for i, (gw, gb) in enumerate(gv2):
        assert relerr(gw, g[f"gv.model.{2 * i}.weight"]) < 1e-5
    # [r3] ... and the POLICY gradients against the fixture's own gp.* tensors, directly: each implementation's gradient minus the
    # contributions of the 32 clip-edge rows under the branch IT took for them (fp64_gate.without_edge_rows) depends on no
    # knife-edge decision -- the HIP kernels and the reference's float32 autograd must agree on it to 1e-5 of every tensor.
    fix_gp = [(torch.as_tensor(g[f"gp.model.{2 * i}.weight"]), torch.as_tensor(g[f"gp.model.{2 * i}.bias"])) for i in range(len(pol))]
    edge, args = out["edge"], ("discrete", pol, val, g["obs"], g["acts"], g["old_logp"], g["adv"], g["targets"], 0.2, 0.005, 0.25, (0.1, 1.0))
    w_fix = fp64_gate._edge_decisions(*args, out["cpu"]["masks"][0], out["cpu"]["masks"][1], fix_gp, edge, 96)
    hip_part = fp64_gate.without_edge_rows(*args, gp2, out["hip"]["masks"], edge, out["hip"]["w"])
    fix_part = fp64_gate.without_edge_rows(*args, fix_gp, out["cpu"]["masks"], edge, w_fix)
    worst = 0.0
    for (hw, hb), (fw, fb) in zip(hip_part, fix_part):
        worst = max(worst, np.abs(hw - fw).max() / np.abs(fw).max(), np.abs(hb - fb).max() / np.abs(fb).max())
    print(f"[fp64 gate] G4 policy gradients without the clip-edge rows' contributions, HIP vs the reference fixture directly: {worst:.2e} "
          f"(branch decisions differing between the two on {int((out['hip']['w'][edge] != w_fix[edge]).sum())} of {len(edge)} edge rows)")
    assert worst < 1e-5, worst


@pytest.mark.parametrize("head,fixture", [("gaussian", "g9_continuous"), ("multidiscrete", "g9_multidiscrete")])
def test_g9_other_heads_minibatch(L, golden, head, fixture):
    g = golden(fixture)
    pol, val = nets.params_from_state(g, "p."), nets.params_from_state(g, "v.")
    n = g["obs"].shape[0]
    acts = g["act"].astype(np.float32)
    got = run_minibatch(L, head, pol, val, g["obs"], acts, g["old_logp"], g["targets"], g["adv"], np.arange(n), 0.2, 0.005, 0.5)
    fp64_gate.gate(L, head, pol, val, g["obs"], acts, g["old_logp"], g["adv"], g["targets"], 0.2, 0.005, 0.5, got, label="G9 " + head)


def test_minibatch_gather_and_accumulate_cfg2_shape(L):
    # 256x3 nets, obs 107, 90 actions; 2 minibatches of 3000 rows gathered from a 10000-row buffer; grads accumulate
    torch.manual_seed(123)
    pol = nets.init_mlp(107, (256, 256, 256), 90)
    val = nets.init_mlp(107, (256, 256, 256), 1)
    rs = np.random.RandomState(1)
    n = 10000
    obs = np.clip(rs.randn(n, 107), -5, 5).astype(np.float32)
    probs = nets.discrete_probs(pol, obs)
    act, logp = nets.discrete_sample(probs, nets.draw_exp_noise(n, 90))
    old = (logp + torch.as_tensor(rs.randn(n).astype(np.float32) * 0.2)).numpy()
    adv = rs.randn(n).astype(np.float32)
    tgt = rs.randn(n).astype(np.float32)
    perm = rs.permutation(n)
    # No row is left out: a hidden unit whose pre-activation is within fp32 GEMM rounding of 0 gets its ReLU mask from the last
    # bit of the accumulation order (CPU-MKL vs MFMA included); the gate reads the masks each implementation took, checks that
    # every one that differs from float64's is such a unit, and compares against float64 under those masks.
    acc_p = acc_v = None
    for s in range(2):
        idx = perm[s * 3000:(s + 1) * 3000]
        got = run_minibatch(L, "discrete", pol, val, obs, act.numpy(), old, tgt, adv, idx, 0.2, 0.005, 0.5)
        fp64_gate.gate(L, "discrete", pol, val, obs[idx], act.numpy()[idx], old[idx], adv[idx], tgt[idx], 0.2, 0.005, 0.5, got,
                       label=f"cfg2 shape, 3000 gathered rows (slice {s})")


def test_fused_gather_and_paired_launches_are_bitwise_neutral(L):
    """[r3] The first layer's four launches fetch their rows straight from the experience arrays through the row table (the
    minibatch gather fused into the GEMMs' load stage), and policy and critic layers of equal widths run as ONE launch
    (csrc/api.hip, paired pass).  Neither changes a product or the order of a sum: gradients must be BIT-identical to the separate
    gather pass (rlppo_dbg_set(26, 0)) and to one launch chain per network (rlppo_dbg_set(29, 0)), on a ragged minibatch (1500
    rows: a partial row tile, a partial dW stage) drawn at random, with repeats, from a 5000-row buffer."""
    rs = np.random.RandomState(21)
    d, A, n, mb = 107, 90, 5000, 1500
    torch.manual_seed(21)
    pol, val = nets.init_mlp(d, (256, 256, 256), A), nets.init_mlp(d, (256, 256, 256), 1)
    obs = np.clip(rs.randn(n, d), -5, 5).astype(np.float32)
    acts = rs.randint(0, A, n).astype(np.float32)
    old = (-np.log(A) + 0.1 * rs.randn(n)).astype(np.float32)
    tgt, adv = rs.randn(n).astype(np.float32), rs.randn(n).astype(np.float32)
    idx = rs.randint(0, n, mb)
    idx[:3] = [n - 1, 0, n - 1]
    runs = {}
    # (26 = 2: the fused gather at every size -- the library's default engages it from 262,144 rows per pass)
    forms = dict(fused=(2, 2, 0, 1), separate_gather=(0, 2, 0, 1), two_chains=(2, 0, 0, 1), round2=(0, 0, 0, 1), stacked_pairs=(2, 2, 0, 0),
                 folded_value_head=(2, 2, 1, 1))
    for key, (k26, k29, k32, k33) in forms.items():
        check(L, L.rlppo_dbg_set(26, k26))
        check(L, L.rlppo_dbg_set(29, k29))   # paired launches (policy + critic layer in one grid) against one chain per network
        check(L, L.rlppo_dbg_set(32, k32))   # the critic's output layer inside the last hidden layer's forward epilogue
        check(L, L.rlppo_dbg_set(33, k33))   # the two products' tiles interleaved in the grid / the second stacked behind the first
        try:
            runs[key] = run_minibatch(L, "discrete", pol, val, obs, acts, old, tgt, adv, idx, 0.2, 0.005, 0.25)
        finally:
            check(L, L.rlppo_dbg_set(26, 1))
            check(L, L.rlppo_dbg_set(29, 1))
            check(L, L.rlppo_dbg_set(32, 1))
            check(L, L.rlppo_dbg_set(33, 1))
    gp0, gv0, st0 = runs["fused"]
    for key, (gp, gv, st) in runs.items():
        if key == "folded_value_head":
            continue
        for (a, b), (c, e) in zip(gp0 + gv0, gp + gv):
            assert torch.equal(a, c) and torch.equal(b, e), key
        # (the report statistics are double atomics of both chains: their order, hence the last bit, may differ)
        np.testing.assert_allclose(st0, st, rtol=1e-12, atol=0, err_msg=key)
    # The folded value head computes each value as two partial dot products over 128 activations each (+ bias) instead of the
    # matrix-vector kernel's order: the policy's gradients cannot change, the critic's only by that rounding -- and the form is
    # deterministic (two partial sums added to zero commute): a second run reproduces it bit for bit.
    gp1, gv1, st1 = runs["folded_value_head"]
    for (a, b), (c, e) in zip(gp0, gp1):
        assert torch.equal(a, c) and torch.equal(b, e)
    for (a, b), (c, e) in zip(gv0, gv1):
        assert relerr(c, a) < 2e-6 and relerr(e, b) < 2e-6
    check(L, L.rlppo_dbg_set(29, 2))
    try:
        gp2, gv2, _ = run_minibatch(L, "discrete", pol, val, obs, acts, old, tgt, adv, idx, 0.2, 0.005, 0.25)
    finally:
        check(L, L.rlppo_dbg_set(29, 1))
    for (a, b), (c, e) in zip(gp1 + gv1, gp2 + gv2):
        assert torch.equal(a, c) and torch.equal(b, e)
    # [r5] every form above ran its weight gradients as ONE grouped launch (the default; the library counts them): the plan of that
    # launch depends on the set of products and the row count only, which is why the forms stay bit-identical.  Against one launch +
    # reduction per layer (rlppo_dbg_set(37, 0)) the gradients differ by the order of their row sums only.
    c5 = L.rlppo_dbg_counter(5)
    assert c5 > 0
    check(L, L.rlppo_dbg_set(37, 0))
    try:
        gp3, gv3, _ = run_minibatch(L, "discrete", pol, val, obs, acts, old, tgt, adv, idx, 0.2, 0.005, 0.25)
    finally:
        check(L, L.rlppo_dbg_set(37, 1))
    assert L.rlppo_dbg_counter(5) == c5
    for (a, b), (c, e) in zip(gp0 + gv0, gp3 + gv3):
        assert relerr(c, a) < 2e-6 and relerr(e, b) < 2e-6
    runs["per_layer_dw"] = (gp3, gv3, st0)
    # and both forms are right (float64 truth), not merely self-consistent
    for key in ("fused", "folded_value_head", "per_layer_dw"):
        fp64_gate.gate(L, "discrete", pol, val, obs[idx], acts[idx], old[idx], adv[idx], tgt[idx], 0.2, 0.005, 0.25, runs[key],
                       label=key + ", ragged 1500-row minibatch")


def test_minibatch_full_size_cfg2(L):
    """BASELINE configs[1] at its real minibatch size: 65,536 rows gathered from a 100,000-row buffer, 256x3 nets.
    (a) additivity, the size-independent property of the update: the gradient of the whole minibatch equals the sum of
        the gradients of its two halves (each scaled by mb_ratio/2) -- different row splits, tile counts and partial-tile
        reductions must agree; the report statistics are means, so they average;
    (b) float64 truth next to the CPU float32 oracle on ALL 65,536 rows (tests/fp64_gate.py)."""
    torch.manual_seed(321)
    pol = nets.init_mlp(107, (256, 256, 256), 90)
    val = nets.init_mlp(107, (256, 256, 256), 1)
    rs = np.random.RandomState(7)
    n = 100000
    obs = np.clip(rs.randn(n, 107), -5, 5).astype(np.float32)
    probs = nets.discrete_probs(pol, obs)
    act, logp = nets.discrete_sample(probs, nets.draw_exp_noise(n, 90))
    old = (logp + torch.as_tensor(rs.randn(n).astype(np.float32) * 0.2)).numpy()
    adv = rs.randn(n).astype(np.float32)
    tgt = rs.randn(n).astype(np.float32)
    perm = rs.permutation(n)
    idx = perm[:65536]
    gp, gv, st = run_minibatch(L, "discrete", pol, val, obs, act.numpy(), old, tgt, adv, idx, 0.2, 0.005, 1.0)
    gp1, gv1, st1 = run_minibatch(L, "discrete", pol, val, obs, act.numpy(), old, tgt, adv, idx[:32768], 0.2, 0.005, 0.5)
    gp2, gv2, st2 = run_minibatch(L, "discrete", pol, val, obs, act.numpy(), old, tgt, adv, idx[32768:], 0.2, 0.005, 0.5)
    for whole, h1, h2 in zip(gp + gv, gp1 + gv1, gp2 + gv2):
        for k in (0, 1):
            assert relerr(whole[k], h1[k].double() + h2[k].double()) < 1e-5
    np.testing.assert_allclose(st[:5], (st1[:5] + st2[:5]) / 2, rtol=1e-5, atol=1e-8)
    fp64_gate.gate(L, "discrete", pol, val, obs[idx], act.numpy()[idx], old[idx], adv[idx], tgt[idx], 0.2, 0.005, 1.0, (gp, gv, st),
                   label="cfg2 full minibatch, 65,536 rows, none excluded")


def test_fused_pass_full_size_cfg2(L):
    """The launch shape of the update at one GPU: ONE pass over the 8 minibatches of a batch = 524,288 rows (BASELINE configs[1]:
    B = 524,288, MB = 65,536; PPOLearner.max_fused_minibatches).  Additivity at that size: the gradient and the report sums of
    the fused pass equal those of its eight 65,536-row minibatches (mb_ratio 1/8 each) -- different row splits, XCD tile
    orders, dW split sizes and partial-tile reductions must agree."""
    torch.manual_seed(99)
    pol = nets.init_mlp(107, (256, 256, 256), 90)
    val = nets.init_mlp(107, (256, 256, 256), 1)
    rs = np.random.RandomState(17)
    n = 524288
    obs = np.clip(rs.randn(n, 107), -5, 5).astype(np.float32)
    with torch.no_grad():
        act, logp = nets.discrete_sample(nets.discrete_probs(pol, obs), nets.draw_exp_noise(n, 90))
    old = (logp + torch.as_tensor(rs.randn(n).astype(np.float32) * 0.2)).numpy()
    adv = rs.randn(n).astype(np.float32)
    tgt = rs.randn(n).astype(np.float32)
    idx = rs.permutation(n)
    gp, gv, st = run_minibatch(L, "discrete", pol, val, obs, act.numpy(), old, tgt, adv, idx, 0.2, 0.005, 1.0)
    acc = [[torch.zeros_like(w, dtype=torch.float64), torch.zeros_like(b, dtype=torch.float64)] for w, b in gp + gv]
    st_sum = np.zeros(5)
    for j in range(8):
        gpj, gvj, stj = run_minibatch(L, "discrete", pol, val, obs, act.numpy(), old, tgt, adv, idx[j * 65536:(j + 1) * 65536], 0.2,
                                      0.005, 0.125)
        for a, g in zip(acc, gpj + gvj):
            a[0] += g[0].double()
            a[1] += g[1].double()
        st_sum += stj[:5]
    for whole, parts in zip(gp + gv, acc):
        for k in (0, 1):
            assert relerr(whole[k], parts[k]) < 1e-5
    np.testing.assert_allclose(st[:5], st_sum / 8, rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("M,N,K", [(256, 256, 32), (1000, 256, 96), (4227, 512, 256), (65536, 256, 256), (140037, 256, 256), (70001, 512, 128)])
def test_gemm_nt_x3_forward_and_dx(L, M, N, K):
    """[r4] The split-bf16 hidden products (csrc/gemm_split.hip; OPT-IN update precision 2): fp32 operands in memory, three bf16
    pieces each, six piece products on the bf16 MFMA pipe, the five small ones summed apart from the accumulator.  Against float64:
    the forward (bias + relu) and the masked dX must be AT LEAST as accurate as the product's fp32-MFMA kernels on the same
    operands (max and rms error, 5 % of slack for the rms); the ReLU bitmask the forward writes is the one every other kernel of
    the update reads -- word for word the fp32 kernel's except where a pre-activation sits within rounding of 0 -- and the dX form
    applies a bitmask written by the fp32 kernel exactly as the fp32 dX kernel does; the packed planes (read back through the
    bank-conflict-free chunk order of their LDS image) add up to the weights (2^-24 relative: the third piece rounds the second residual).  Ragged row counts, two column tiles, K = 32 ... 256."""
    g = torch.Generator(device="cuda").manual_seed(M + K)
    A = (torch.randn(M, K, device="cuda", generator=g).clamp_(min=0) * torch.rand(M, K, device="cuda", generator=g)).contiguous()
    W = ((torch.rand(N, K, device="cuda", generator=g) * 2 - 1) / np.sqrt(K)).contiguous()
    bias = ((torch.rand(N, device="cuda", generator=g) - 0.5) * 0.2).contiguous()
    planes = torch.zeros(3 * N * K, dtype=torch.bfloat16, device="cuda")
    check(L, L.rlppo_dbg_pack_x3(stream(), P(W), K, N, K, P(planes)))
    # a plane's 16-column block is 64 chunks of 8 k: (column r, k quarter q) at chunk r * 4 + (q ^ (-(r // 4) & 3))  (wpos(), gemm_split.hip)
    r16, q4 = torch.arange(16).view(16, 1), torch.arange(4).view(1, 4)
    pos = (r16 * 4 + (q4 ^ ((-(r16 // 4)) & 3))).reshape(-1).cuda()
    pl = planes.view(K // 32, 3, N // 16, 64, 8).float()[:, :, :, pos].reshape(K // 32, 3, N, 32)
    rebuilt = (pl[:, 0] + pl[:, 1] + pl[:, 2]).permute(1, 0, 2).reshape(N, K)
    assert (rebuilt - W).abs().max().item() <= 2.0 ** -24 * W.abs().max().item()
    nb = max(int(L.rlppo_dbg_gemm_nt_bits_bytes(M, N)), 8)
    bits32, bits3 = torch.zeros(nb, dtype=torch.uint8, device="cuda"), torch.zeros(nb, dtype=torch.uint8, device="cuda")
    C32, C3 = torch.empty(M, N, device="cuda"), torch.full((M, N), float("nan"), device="cuda")
    check(L, L.rlppo_dbg_gemm_nt_bits(stream(), P(A), K, P(W), K, P(bias), P(C32), N, M, N, K, 1, P(bits32)))
    check(L, L.rlppo_dbg_gemm_nt_x3(stream(), P(A), K, P(planes), P(bias), P(C3), N, M, N, K, 0, P(bits3)))
    rows = torch.unique(torch.cat([torch.arange(0, M, max(1, M // 2048), device="cuda"), torch.arange(max(0, M - 300), M, device="cuda")]))
    pre = A[rows].double() @ W.double().t() + bias.double()
    truth = torch.relu(pre)
    scale = truth.abs().max().item()
    e32, e3 = (C32[rows].double() - truth).abs(), (C3[rows].double() - truth).abs()
    print(f"[x3] forward {M}x{N}x{K}: err vs float64 (of max|C|) split-bf16 max {e3.max().item() / scale:.2e} rms {e3.pow(2).mean().sqrt().item() / scale:.2e}; "
          f"fp32 MFMA max {e32.max().item() / scale:.2e} rms {e32.pow(2).mean().sqrt().item() / scale:.2e}")
    assert torch.isfinite(C3).all()
    assert e3.max().item() <= max(1.0 * e32.max().item(), 2e-7 * scale) and e3.pow(2).mean().sqrt().item() <= 1.05 * e32.pow(2).mean().sqrt().item() + 1e-9 * scale
    # bitmask: the same words as the fp32 kernel's but for pre-activations within rounding of zero
    diff = (bits32 != bits3)
    n_diff_bits = int(sum(bin(int(x)).count("1") for x in (bits32[diff] ^ bits3[diff]).cpu().numpy()))
    near_zero = int((pre.abs() < 1e-6 * scale).sum().item()) * max(1, M // rows.numel()) + 8
    assert n_diff_bits <= near_zero, (n_diff_bits, near_zero)
    # dX through the FP32 kernel's bitmask: dX[M][K'] = (dY[M][N'] . Wd[N'][K']) masked; here dY = C32 (N' = N), K' = N2
    N2 = 256
    Wd = ((torch.rand(N2, N, device="cuda", generator=g) * 2 - 1) / np.sqrt(N)).contiguous()   # B operand [out cols N2][contraction N]
    planes_d = torch.zeros(3 * N2 * N, dtype=torch.bfloat16, device="cuda")
    check(L, L.rlppo_dbg_pack_x3(stream(), P(Wd), N, N2, N, P(planes_d)))
    dY = torch.randn(M, N, device="cuda", generator=g)
    mask_src = torch.randn(M, N2, device="cuda", generator=g)
    nb2 = max(int(L.rlppo_dbg_gemm_nt_bits_bytes(M, N2)), 8)
    mbits = torch.zeros(nb2, dtype=torch.uint8, device="cuda")
    scratch = torch.empty(M, N2, device="cuda")
    zero_b = torch.zeros(N2, device="cuda")
    # a bitmask in the library's layout for `mask_src > 0`: the fp32 forward kernel with identity weights would need K = N2; instead let
    # it write the mask of relu(mask_src . I + 0) -- the fp32 kernel, K = N2
    eye = torch.eye(N2, device="cuda").contiguous()
    check(L, L.rlppo_dbg_gemm_nt_bits(stream(), P(mask_src), N2, P(eye), N2, P(zero_b), P(scratch), N2, M, N2, N2, 1, P(mbits)))
    dX32, dX3 = torch.empty(M, N2, device="cuda"), torch.full((M, N2), float("nan"), device="cuda")
    check(L, L.rlppo_dbg_gemm_nt_bits(stream(), P(dY), N, P(Wd), N, None, P(dX32), N2, M, N2, N, 3, P(mbits)))
    check(L, L.rlppo_dbg_gemm_nt_x3(stream(), P(dY), N, P(planes_d), None, P(dX3), N2, M, N2, N, 1, P(mbits)))
    tr = (dY[rows].double() @ Wd.double().t()) * (mask_src[rows] > 0)
    sc = tr.abs().max().item()
    d32, d3 = (dX32[rows].double() - tr).abs(), (dX3[rows].double() - tr).abs()
    print(f"[x3] dX {M}x{N2}x{N}: split-bf16 max {d3.max().item() / sc:.2e} rms {d3.pow(2).mean().sqrt().item() / sc:.2e}; fp32 MFMA max {d32.max().item() / sc:.2e} rms {d32.pow(2).mean().sqrt().item() / sc:.2e}")
    assert torch.equal(dX3 == 0, dX32 == 0) or ((dX3 == 0) != (dX32 == 0)).sum().item() <= 4   # the same entries masked
    assert d3.max().item() <= max(1.0 * d32.max().item(), 2e-7 * sc) and d3.pow(2).mean().sqrt().item() <= 1.05 * d32.pow(2).mean().sqrt().item() + 1e-9 * sc


def test_gemm_nt_x3_special_values(L):
    """[r5] The split-bf16 products beside the fp32 MFMA kernel on operands the N(0, 1) tests never draw (include/rlppo.h states
    the behaviour this pins): rows of 1e-30-scale and of denormal activations stay as accurate, relative to their own magnitude,
    as the fp32 kernel's; rows WITHOUT a special value are bit-identical to the run without any (a special value does not leak
    beyond its row); a row holding +-inf, NaN or a finite |x| >= 3.396e38 (which rounds to a bf16 infinity) comes out NaN over the
    whole row -- the documented deviation: x - bf16(x) is inf - inf there, where the fp32 MFMA's fmaf chain returns +-inf (or, for
    the huge finite value, a finite product)."""
    M, N, K = 512, 256, 256
    g = torch.Generator(device="cuda").manual_seed(99)
    A = torch.randn(M, K, device="cuda", generator=g).contiguous()
    W = ((torch.rand(N, K, device="cuda", generator=g) * 2 - 1) / np.sqrt(K)).contiguous()
    planes = torch.zeros(3 * N * K, dtype=torch.bfloat16, device="cuda")
    check(L, L.rlppo_dbg_pack_x3(stream(), P(W), K, N, K, P(planes)))
    nb = max(int(L.rlppo_dbg_gemm_nt_bits_bytes(M, N)), 8)
    ones = torch.full((nb,), 0xFF, dtype=torch.uint8, device="cuda")           # a mask that keeps everything: the product itself (mode 1 = dX)

    def both(a):
        c32, c3 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
        check(L, L.rlppo_dbg_gemm_nt_bits(stream(), P(a), K, P(W), K, None, P(c32), N, M, N, K, 3, P(ones)))
        check(L, L.rlppo_dbg_gemm_nt_x3(stream(), P(a), K, P(planes), None, P(c3), N, M, N, K, 1, P(ones)))
        return c32, c3

    base32, base3 = both(A)
    S = A.clone()
    special = {3: float("inf"), 10: float("nan"), 20: 3.4e38, 30: float("-inf")}
    for r, v in special.items():
        S[r, 5] = v
    S[40:48] *= 1e-30
    S[50:58] *= 1e-38                                                           # fp32 denormals (|x| < 1.18e-38) among them
    c32, c3 = both(S)
    plain = torch.ones(M, dtype=torch.bool, device="cuda")
    plain[list(special)] = False
    plain[40:48] = plain[50:58] = False
    assert torch.equal(c3[plain], base3[plain]) and torch.equal(c32[plain], base32[plain])   # no leak beyond the row
    truth = S.double() @ W.double().t()
    for lo, hi, tag in ((40, 48, "1e-30 scale"), (50, 58, "1e-38 scale (denormal operands)")):
        sc = truth[lo:hi].abs().max().item()
        e32, e3 = (c32[lo:hi].double() - truth[lo:hi]).abs().max().item() / sc, (c3[lo:hi].double() - truth[lo:hi]).abs().max().item() / sc
        print(f"[x3 special] {tag}: err / max|C| of the rows: split-bf16 {e3:.2e}, fp32 MFMA {e32:.2e}")
        if lo == 40:
            assert e3 <= max(e32, 2e-7)         # normal fp32 range: the pieces are normal bf16 numbers, nothing is lost
        else:
            assert e3 <= 1e-2                    # pieces below bf16's normal range (2^-126) lose their low bits: relative to 1e-38, not to the data
    for r, v in special.items():
        print(f"[x3 special] row {r} holds {v}: fp32 MFMA -> {int(torch.isinf(c32[r]).sum())} inf / {int(torch.isnan(c32[r]).sum())} NaN / "
              f"{int(torch.isfinite(c32[r]).sum())} finite; split-bf16 -> {int(torch.isinf(c3[r]).sum())} inf / {int(torch.isnan(c3[r]).sum())} NaN")
        assert torch.isnan(c3[r]).all()                                          # the documented deviation (never a finite wrong number)
        assert not torch.isfinite(c32[r]).all() or v == 3.4e38


def test_minibatch_x3_precision_cfg2_shape(L):
    """[r4] rlppo_ppo_minibatch in the split-bf16 update precision at the cfg2 shape (256x3 nets, obs 107, 90 actions; hidden
    forwards of layers 1-2, their dX and the head's dX on the split kernels), 3000 gathered rows and the full 65,536-row
    minibatch: through the SAME float64 gate as the fp32 precision (err(HIP, fp64) <= max(1e-5, 1.5 x the CPU fp32 oracle's),
    masks read through the split kernel), and the gradients within fp32 rounding noise of the fp32 precision's own."""
    torch.manual_seed(123)
    pol = nets.init_mlp(107, (256, 256, 256), 90)
    val = nets.init_mlp(107, (256, 256, 256), 1)
    rs = np.random.RandomState(1)
    n = 70000
    obs = np.clip(rs.randn(n, 107), -5, 5).astype(np.float32)
    probs = nets.discrete_probs(pol, obs)
    act, logp = nets.discrete_sample(probs, nets.draw_exp_noise(n, 90))
    old = (logp + torch.as_tensor(rs.randn(n).astype(np.float32) * 0.2)).numpy()
    adv = rs.randn(n).astype(np.float32)
    tgt = rs.randn(n).astype(np.float32)
    perm = rs.permutation(n)
    for rows, ratio in ((3000, 0.5), (65536, 1.0)):
        idx = perm[:rows]
        got3 = run_minibatch(L, "discrete", pol, val, obs, act.numpy(), old, tgt, adv, idx, 0.2, 0.005, ratio, precision="x3")
        got32 = run_minibatch(L, "discrete", pol, val, obs, act.numpy(), old, tgt, adv, idx, 0.2, 0.005, ratio)
        fp64_gate.gate(L, "discrete", pol, val, obs[idx], act.numpy()[idx], old[idx], adv[idx], tgt[idx], 0.2, 0.005, ratio, got3,
                       label=f"cfg2 shape, {rows} rows, split-bf16 update precision", x3=True)
        worst = max(max(relerr(a[0], b[0]), relerr(a[1], b[1])) for a, b in zip(got3[0] + got3[1], got32[0] + got32[1]))
        print(f"[x3] {rows} rows: gradients of the split-bf16 precision vs the fp32 precision's: {worst:.2e} of max|g| per tensor")
        assert worst < 2e-3   # (a ReLU decision within rounding of 0 may differ between the two: O(1/sqrt(rows)) of a first-layer row)


def test_learn_report_entry_point(L):
    """[r6] rlppo_learn_report through the C ABI alone (ppo_learner.py:213-236: the update magnitudes, the report sums read out): the
    two norms against float64 of the float32 differences (any sizes, incl. lengths that are no multiple of anything), the accumulators
    copied out with add_passes added and ZEROED, the give-up word and the extra double passed through, the completion word released
    with the call's value; the ticket block re-arms itself (five calls on one block), results are bit-identical from call to call."""
    from rlgym_ppo_amd import _native as N
    torch.manual_seed(5)
    ws = torch.zeros(N.REPORT_WS_BYTES, dtype=torch.uint8, device="cuda")
    out = torch.zeros(N.REPORT_OUT_DOUBLES, dtype=torch.float64).pin_memory()
    done = torch.zeros(1, dtype=torch.int32).pin_memory()
    word = torch.tensor([3], dtype=torch.int32, device="cuda")
    extra = torch.tensor([11.0], dtype=torch.float64, device="cuda")
    seen = []
    for call, (n_pol, n_val) in enumerate(((182362, 159489), (1, 0), (70001, 13), (182362, 159489), (182362, 159489))):
        g = torch.Generator(device="cuda").manual_seed(100 if call >= 3 else call)
        pb, pn = torch.randn(n_pol, device="cuda", generator=g), torch.randn(n_pol, device="cuda", generator=g)
        vb, vn = torch.randn(max(n_val, 1), device="cuda", generator=g), torch.randn(max(n_val, 1), device="cuda", generator=g)
        stats = torch.arange(1, N.N_STATS + 1, dtype=torch.float64, device="cuda") * 0.5
        a = N.ReportArgs()
        a.pol_before, a.pol_now, a.n_pol, a.val_before, a.val_now, a.n_val = pb.data_ptr(), pn.data_ptr(), n_pol, vb.data_ptr(), vn.data_ptr(), n_val
        a.stats, a.add_passes = stats.data_ptr(), 4.0
        a.timeout_word, a.extra = (word.data_ptr(), extra.data_ptr()) if call % 2 == 0 else (None, None)
        a.out, a.done_word, a.done_value, a.ws = out.data_ptr(), done.data_ptr(), 1000 + call, ws.data_ptr()
        done.zero_()
        check(L, L.rlppo_learn_report(stream(), ctypes.byref(a)))
        assert L.rlppo_host_wait_words(done.data_ptr(), 1, 1000 + call, 2_000_000) == 0
        got = out.numpy().copy()
        want = (np.arange(1, N.N_STATS + 1) * 0.5)
        want[N.STAT_PASSES] += 4.0
        assert np.array_equal(got[:N.N_STATS], want) and float(stats.abs().sum()) == 0.0
        for k, (b, n_, cnt) in enumerate(((pb, pn, n_pol), (vb, vn, n_val))):
            ref = float((b[:cnt] - n_[:cnt]).double().norm()) if cnt else 0.0
            assert abs(got[N.N_STATS + k] - ref) <= 1e-12 * max(ref, 1.0), (call, k, got[N.N_STATS + k], ref)
        assert got[N.N_STATS + 2] == (3.0 if call % 2 == 0 else 0.0) and got[N.N_STATS + 3] == (11.0 if call % 2 == 0 else 0.0)
        seen.append(got[N.N_STATS:N.N_STATS + 2].copy())
    assert np.array_equal(seen[3], seen[4])      # the same inputs, the same bits (fixed-order sums)
    assert int(ws[:4].view(torch.int32).item()) == 0   # the ticket counter is back at zero
    # argument errors come back as codes
    bad = N.ReportArgs()
    assert L.rlppo_learn_report(stream(), ctypes.byref(bad)) == 1001


def test_clip_adam_matches_oracle(L):
    torch.manual_seed(0)
    params = nets.init_mlp(20, (16,), 5)
    st = ppo.AdamState(params)
    flat = dev(nets.flatten(params))
    m = torch.zeros_like(flat)
    v = torch.zeros_like(flat)
    gn = torch.zeros(1, dtype=torch.float64, device="cuda")
    for step in range(1, 6):
        grads = [(torch.randn_like(w) * (3.0 if step % 2 else 0.01), torch.randn_like(b)) for w, b in params]
        gflat = dev(nets.flatten(grads))
        check(L, L.rlppo_clip_adam(stream(), P(flat), P(gflat), P(m), P(v), flat.numel(), 0.5, 3e-4, 0.9, 0.999, 1e-8, step, P(gn)))
        coef, total = ppo.clip_coef(grads)
        assert abs(float(gn.cpu()) ** 0.5 - float(total)) < 1e-5 * float(total)
        ppo.adam_step(params, [(w * coef, b * coef) for w, b in grads], st, 3e-4)
        assert relerr(flat, nets.flatten(params)) < 1e-6
        assert relerr(m, nets.flatten(st.m)) < 1e-5 and relerr(v, nets.flatten(st.v)) < 1e-5
        assert relerr(gflat, nets.flatten([(w * coef, b * coef) for w, b in grads])) < 1e-6  # grads scaled in place


@pytest.mark.parametrize("one_launch", [True, False], ids=["one_launch_grid_barrier", "three_operations"])
def test_clip_adam_pack2_equals_separate_launches(L, one_launch):
    """rlppo_clip_adam_pack2 (both nets' clip + Adam + re-pack + zero_grad; ONE launch with a grid barrier between the norms and the
    update when the caller hands over a sync block -- what PPOLearner does -- else fill + norms + update) against rlppo_clip_adam x2 +
    rlppo_net_pack x2: parameters, both Adam moments and the packed copies bit-identical over several steps (large and tiny
    gradients: clipped and unclipped), gradients left zero, squared norms equal; odd layer widths exercise the padding."""
    from rlgym_ppo_amd import _native as N
    torch.manual_seed(3)
    shapes = [[107, 256, 256, 90], [107, 64, 1]]
    state = []
    for dims in shapes:
        dc = N.dims_array(dims)
        nl = len(dims) - 1
        nf, npk = int(L.rlppo_flat_floats(dc, nl)), int(L.rlppo_packed_floats(dc, nl))
        flat = torch.randn(nf, device="cuda") * 0.1
        st = dict(dc=dc, nl=nl, nf=nf)
        for tag in ("a", "b"):  # a: separate launches, b: fused
            st[tag] = dict(p=flat.clone(), m=torch.zeros(nf, device="cuda"), v=torch.zeros(nf, device="cuda"),
                           packed=torch.zeros(npk, device="cuda"), gn=torch.zeros(1, dtype=torch.float64, device="cuda"))
            check(L, L.rlppo_net_pack(stream(), dc, nl, P(st[tag]["p"]), P(st[tag]["packed"])))
        state.append(st)
    sync = torch.zeros(N.OPT_SYNC_BYTES // 4, dtype=torch.int32, device="cuda") if one_launch else None   # zeroed ONCE
    for step in range(1, 6):
        descs = []
        for st in state:
            g = torch.randn(st["nf"], device="cuda") * (5.0 if step % 2 else 1e-4)
            ga, gb = g.clone(), g.clone()
            a, b = st["a"], st["b"]
            check(L, L.rlppo_clip_adam(stream(), P(a["p"]), P(ga), P(a["m"]), P(a["v"]), st["nf"], 0.5, 3e-4, 0.9, 0.999, 1e-8, step, P(a["gn"])))
            check(L, L.rlppo_net_pack(stream(), st["dc"], st["nl"], P(a["p"]), P(a["packed"])))
            d = N.OptNet()
            d.dims, d.n_layers = ctypes.cast(st["dc"], ctypes.POINTER(ctypes.c_int32)), st["nl"]
            d.params, d.grads, d.exp_avg, d.exp_avg_sq = b["p"].data_ptr(), gb.data_ptr(), b["m"].data_ptr(), b["v"].data_ptr()
            d.packed, d.gnorm2 = b["packed"].data_ptr(), b["gn"].data_ptr()
            d.max_norm, d.lr, d.beta1, d.beta2, d.eps, d.step = 0.5, 3e-4, 0.9, 0.999, 1e-8, step
            descs.append((d, gb))
        check(L, L.rlppo_clip_adam_pack2(stream(), ctypes.byref(descs[0][0]), ctypes.byref(descs[1][0]), P(sync)))
        torch.cuda.synchronize()
        for st, (_, gb) in zip(state, descs):
            a, b = st["a"], st["b"]
            for k in ("p", "m", "v", "packed"):
                assert torch.equal(a[k], b[k]), (step, k)
            assert (gb == 0).all()
            assert abs(a["gn"].item() - b["gn"].item()) <= 1e-12 * a["gn"].item()
    if one_launch:  # the barrier leaves its block armed (accumulators and counter back at zero) and never timed out
        w = sync.cpu().numpy()   # header: arrival counter, generation, timeouts
        assert w[0] == 0 and w[1] == 5 and w[2] == 0, w[:4]
    bad = N.OptNet()
    assert L.rlppo_clip_adam_pack2(stream(), ctypes.byref(bad), ctypes.byref(descs[1][0]), P(sync)) != 0
    if not one_launch:
        return
    # A barrier wait that gives up is ALL OR NOTHING (advisor finding, round 3: it used to write NaN).  rlppo_dbg_set(35, 1) keeps one
    # workgroup from ever arriving (a grid that is not co-resident), rlppo_dbg_set(34, 0) makes the waiters give up at once: no
    # element of either network changes, gradients included; the block counts the event and is dead -- the next call on it skips
    # too -- until its owner zeroes it; after that, and with the hooks off, the same step goes through and equals the
    # three-operation form's.
    snap = lambda: [[st["b"][k].clone() for k in ("p", "m", "v", "packed")] for st in state]
    for (d, gb), st in zip(descs, state):
        gb.copy_(torch.randn(st["nf"], device="cuda"))
        d.step = 6
    before, g_before = snap(), [gb.clone() for _, gb in descs]
    check(L, L.rlppo_dbg_set(34, 0))
    check(L, L.rlppo_dbg_set(35, 1))
    try:
        check(L, L.rlppo_clip_adam_pack2(stream(), ctypes.byref(descs[0][0]), ctypes.byref(descs[1][0]), P(sync)))
        torch.cuda.synchronize()
    finally:
        check(L, L.rlppo_dbg_set(35, 0))
        check(L, L.rlppo_dbg_set(34, -1))
    w = sync.cpu().numpy()
    assert w[2] >= 1 and np.uint32(w[1]) == np.uint32(0xFFFFFFFF), w[:4]
    for b0, b1 in zip(before, snap()):
        assert all(torch.equal(x, y) for x, y in zip(b0, b1))
    assert all(torch.equal(g0, gb) for g0, (_, gb) in zip(g_before, descs))
    assert all(torch.isfinite(st["b"]["p"]).all() for st in state)
    check(L, L.rlppo_clip_adam_pack2(stream(), ctypes.byref(descs[0][0]), ctypes.byref(descs[1][0]), P(sync)))   # dead block: skips at once
    torch.cuda.synchronize()
    assert int(sync[2].item()) > int(w[2])
    for b0, b1 in zip(before, snap()):
        assert all(torch.equal(x, y) for x, y in zip(b0, b1))
    sync.zero_()                                                                                                 # the owner re-arms it
    ref = []
    for (d, gb), st in zip(descs, state):  # the three-operation form on copies = what the repeated step must give
        c = {k: st["b"][k].clone() for k in ("p", "m", "v", "packed")}
        gc, gn = gb.clone(), torch.zeros(1, dtype=torch.float64, device="cuda")
        d2 = N.OptNet()
        d2.dims, d2.n_layers = d.dims, d.n_layers
        d2.params, d2.grads, d2.exp_avg, d2.exp_avg_sq = c["p"].data_ptr(), gc.data_ptr(), c["m"].data_ptr(), c["v"].data_ptr()
        d2.packed, d2.gnorm2 = c["packed"].data_ptr(), gn.data_ptr()
        d2.max_norm, d2.lr, d2.beta1, d2.beta2, d2.eps, d2.step = 0.5, 3e-4, 0.9, 0.999, 1e-8, 6
        ref.append((d2, c, gc, gn))
    check(L, L.rlppo_clip_adam_pack2(stream(), ctypes.byref(ref[0][0]), ctypes.byref(ref[1][0]), None))
    check(L, L.rlppo_clip_adam_pack2(stream(), ctypes.byref(descs[0][0]), ctypes.byref(descs[1][0]), P(sync)))
    torch.cuda.synchronize()
    for st, (_, c, _, _), (_, gb) in zip(state, ref, descs):
        for k in ("p", "m", "v", "packed"):
            assert torch.equal(st["b"][k], c[k]), k
        assert (gb == 0).all()
    assert int(sync[2].item()) == 0
