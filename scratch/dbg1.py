import sys, ctypes, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from rlgym_ppo_amd import _native as N
from test_gpu_kernels import dev, P, stream, run_gae, synth_gae
from oracle import gae as ogae
L = N.lib()
# categorical
torch.manual_seed(3)
probs = torch.softmax(torch.randn(8, 90) * 3, -1).clamp(1e-11, 1)
q = torch.empty(8, 90).exponential_(1)
ref = torch.argmax(probs / q, -1)
act = torch.full((8,), -7, dtype=torch.int64, device="cuda"); lp = torch.full((8,), -7.0, device="cuda")
rc = L.rlppo_categorical_select(stream(), P(dev(probs)), 90, 8, 90, P(dev(q)), P(act), P(lp)); torch.cuda.synchronize()
print("cat rc", rc, act.cpu().tolist(), ref.tolist(), lp.cpu().tolist()[:3], torch.log(probs[torch.arange(8), ref]).tolist()[:3])
# gae small
for n in (1, 7, 64, 300, 5000):
    rs = np.random.RandomState(n); rews = rs.randn(n).astype(np.float32); values = rs.randn(n+1).astype(np.float32)
    dones = (rs.rand(n) < 0.1).astype(np.float32); trunc = np.zeros(n, np.float32)
    vt, adv, ret = run_gae(L, rews, dones, trunc, values, 0.99, 0.95, None)
    ovt, oadv, oret = ogae.gae(rews, dones, trunc, values, 0.99, 0.95, None, "f64")
    bad = np.where(np.abs(adv - oadv) > 1e-4)[0]
    print("gae n", n, "maxerr adv", np.abs(adv-oadv).max(), "ret", np.abs(ret-oret).max(), "nbad", len(bad), bad[:10], adv[:4], oadv[:4])
# gemm_tn
for (M,out,in_) in [(33,90,256),(1,21,32),(64,128,128),(32,16,32)]:
    g = torch.Generator().manual_seed(1)
    ny, kx = int(L.rlppo_padded_out(out)), int(L.rlppo_padded_out(in_))
    dY = torch.zeros(M, ny); dY[:, :out] = torch.randn(M, out, generator=g)
    X = torch.zeros(M, kx); X[:, :in_] = torch.randn(M, in_, generator=g)
    dW = torch.zeros(out, in_, device="cuda"); db = torch.zeros(out, device="cuda")
    rc = L.rlppo_dbg_gemm_tn(stream(), P(dev(dY)), ny, ny, P(dev(X)), kx, None, kx, P(dW), P(db), out, in_, M); torch.cuda.synchronize()
    refW = dY[:, :out].double().T @ X[:, :in_].double(); refb = dY[:, :out].double().sum(0)
    e = (dW.cpu().double() - refW).abs()
    print("tn", M, out, in_, "rc", rc, "maxerr", e.max().item(), "refmax", refW.abs().max().item(), "db err", (db.cpu().double()-refb).abs().max().item())
    bad = (e > 1e-3).nonzero()
    print("   nbad", len(bad), bad[:6].tolist(), "ratio", (dW.cpu().double()/refW)[:2,:4].tolist())
