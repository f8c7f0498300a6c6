"""Turns the FETCH_SIZE / WRITE_SIZE summaries (tools/pmc_summary.py output) into profiles/<tag>_traffic.json:
HBM-side bytes per launch per kernel, with the gfx950 correction of MI355X_MICROARCH.md "HBM": FETCH_SIZE (KiB) counts
128-byte read requests at 64 bytes for wide coalesced streams -> doubled; WRITE_SIZE (KiB) is exact for 16-byte stores
and float atomics.  [r6] The file is stamped with rlppo_build_id() of the library the passes ran on ("_build_id"): bench.py replays
these numbers only while that is the library it has loaded."""
import csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fetch, write, out_path = sys.argv[1:4]
def load(path, col):
    return {r["kernel"]: float(r[col]) for r in csv.DictReader(open(path))}
f, w = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
out = {k: {"fetch_bytes": f[k] * 1024 * 2, "write_bytes": w.get(k, 0.0) * 1024, "hbm_bytes": f[k] * 2048 + w.get(k, 0.0) * 1024,
           "correction": "FETCH_SIZE x2 (gfx950 wide-read under-count), WRITE_SIZE x1"} for k in f}
print(json.dumps({k: round(v["hbm_bytes"] / 1e6, 1) for k, v in out.items()}))
try:
    from rlgym_ppo_amd import _native as N
    out["_build_id"] = N.lib().rlppo_build_id().decode()
except Exception as ex:  # noqa: BLE001
    out["_build_id"] = None
    print("no build id:", ex, file=sys.stderr)
json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
