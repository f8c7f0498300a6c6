"""kernel_breakdown of bench.py at a chosen number of rows per pass (default 65,536 = the pass of one rank of an 8-rank job):
isolated launch time of every GEMM shape of the update, and their sum per pass.  usage: python tools/breakdown_rows.py [rows]"""
import contextlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

with contextlib.redirect_stdout(sys.stderr):
    learner, buf = bench.build_workload("cuda:0")
for M in ([int(x) for x in sys.argv[1:]] or [65536, 524288]):
    learner._fused_rows = M
    rows, dom, _ = bench.kernel_breakdown(learner)
    tot = sum(r["n"] * r["ms"] for r in rows)
    print(f"rows per pass {M}:")
    for r in rows:
        print("  %-46s x%d  %8.4f ms  %7.2f TFLOP/s algorithmic (%.3f of peak)" % (r["kernel"], r["n"], r["ms"], r["tflops"], r["frac"]))
    print("  GEMM launches of one pass, isolated, summed: %.3f ms  (%.3f ms per 65,536 rows)" % (tot, tot * 65536 / M))
