"""A/B of two builds of librlppo (compile-time variants) on the headline workload: alternating processes, each `bench.py --no-extras`.
usage: python tools/ab_lib.py path/to/variant.so [rounds]"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variant = os.path.abspath(sys.argv[1])
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
res = {"default": [], "variant": []}
for r in range(rounds):
    for name, lib in (("default", None), ("variant", variant)):
        env = dict(os.environ)
        if lib:
            env["RLPPO_LIB"] = lib
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "2", "--no-extras"], env=env,
                             capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        res[name].append(json.loads(line)["ms_per_step"])
        print(name, res[name][-1], flush=True)
for k, v in res.items():
    print("%-8s median %.3f ms per step  (%s)" % (k, sorted(v)[len(v) // 2], ", ".join("%.2f" % x for x in v)))
