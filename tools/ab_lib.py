"""A/B of two builds of librlppo (compile-time variants), alternating processes.  Default workload: the headline `bench.py --no-extras`
(ms per step); with `share8`: tools/rank_share.py 8 (ms per learn() of the 8-rank share).
usage: python tools/ab_lib.py path/to/variant.so [rounds] [share8]"""
import json, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variant = os.path.abspath(sys.argv[1])
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
share8 = len(sys.argv) > 3 and sys.argv[3] == "share8"
res = {"default": [], "variant": []}
for r in range(rounds):
    for name, lib in (("default", None), ("variant", variant)):
        env = dict(os.environ)
        if lib:
            env["RLPPO_LIB"] = lib
        if share8:
            out = subprocess.run([sys.executable, os.path.join(root, "tools", "rank_share.py"), "8"], env=env, capture_output=True, text=True, timeout=600)
            res[name].append(float(re.search(r"of 8:\s+([0-9.]+) ms", out.stdout).group(1)))
        else:
            out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "2", "--no-extras"], env=env,
                                 capture_output=True, text=True, timeout=600)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
            res[name].append(json.loads(line)["ms_per_step"])
        print(name, res[name][-1], flush=True)
for k, v in res.items():
    print("%-8s median %.3f ms  (%s)" % (k, sorted(v)[len(v) // 2], ", ".join("%.2f" % x for x in v)))
