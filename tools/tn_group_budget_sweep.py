"""The grouped weight-gradient launch (csrc/gemm.hip: launch_gemm_tn_group) at the update's product set against its workgroup budget
(rlppo_dbg_set(38, n): one round of n workgroups, splits as long as that takes; 0 = the library's own plan: two per CU per round, a
split at most 8192 rows long).  Interleaved, HIP events.  usage: python tools/tn_group_budget_sweep.py [rows ...]"""
import contextlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rlgym_ppo_amd import _native as N

L = N.lib()
with contextlib.redirect_stdout(sys.stderr):
    learner, _ = bench.build_workload("cuda:0")
for M in ([int(x) for x in sys.argv[1:]] or [524288, 65536]):
    learner._fused_rows = M
    res = {}
    budgets = (0, 1024, 2048, 3072, 4096, 6144) if M > 131072 else (0, 512, 1024, 1536, 2048, 3072)
    for rnd in range(3):
        for b in budgets:
            N.check(L.rlppo_dbg_set(38, b))
            rows, _, _ = bench.kernel_breakdown(learner, only="dW all")
            res.setdefault(b, []).append(rows[0]["ms"])
    N.check(L.rlppo_dbg_set(38, 0))
    print("rows per pass %d: grouped dW + reduce, ms by workgroup budget: " % M + ", ".join("%s %.4f" % (b or "plan", float(np.median(v))) for b, v in res.items()))
