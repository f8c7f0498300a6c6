"""Condenses one tools/round_profile.sh session (gpurun_out/<session>/) into the tracked evidence files profiles/<tag>_*:
usage  python tools/collect_profiles.py <session> <tag>   e.g.  r02a r02"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
session, tag = sys.argv[1], sys.argv[2]
src, dst = os.path.join(ROOT, "gpurun_out", session), os.path.join(ROOT, "profiles")


def newest(pattern):
    files = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)
    if not files:
        raise SystemExit("missing " + pattern)
    return files[-1]


copies = {
    "bench.json": f"{tag}_bench.json", "bench_cfg5_fp32.json": f"{tag}_bench_cfg5_fp32.json", "bench_cfg5_bf16.json": f"{tag}_bench_cfg5_bf16.json",
    "pmc_sq.csv": f"{tag}_pmc_sq.csv", "pmc_fetch.csv": f"{tag}_pmc_fetch.csv", "pmc_write.csv": f"{tag}_pmc_write.csv",
    "traffic.json": f"{tag}_traffic.json", "gae_pmc_fetch.csv": f"{tag}_gae_pmc_fetch.csv", "gae_pmc_write.csv": f"{tag}_gae_pmc_write.csv",
    "gae_traffic.json": f"{tag}_gae_traffic.json",
    "cfg5_pmc_fetch.csv": f"{tag}_cfg5_bf16_pmc_fetch.csv", "cfg5_pmc_write.csv": f"{tag}_cfg5_bf16_pmc_write.csv",
    "cfg5_pmc_sq.csv": f"{tag}_cfg5_bf16_pmc_sq.csv", "cfg5_traffic.json": f"{tag}_cfg5_bf16_traffic.json",
    "rank_share.txt": f"{tag}_rank_share.txt", "gae_floor.txt": f"{tag}_gae_floor.txt", "ab_update.txt": f"{tag}_ab_update_final.txt",
    "breakdown_rows.txt": f"{tag}_breakdown_rows.txt", "act_kernel_time.txt": f"{tag}_act_kernel_time.txt",
    "rollout_breakdown.txt": f"{tag}_rollout_breakdown.txt", "small_batch_latency.txt": f"{tag}_small_batch_latency.txt", "get_action_profile.txt": f"{tag}_get_action_profile.txt", "get_action_modes.txt": f"{tag}_get_action_modes.txt", "heads_latency.txt": f"{tag}_heads_latency.txt", "host_window_probe.txt": f"{tag}_host_window_probe.txt", "host_noise_pipeline.txt": f"{tag}_host_noise_pipeline.txt", "fused_act_depth.txt": f"{tag}_fused_act_depth.txt", "rank_share_step_gaps.txt": f"{tag}_rank_share_step_gaps.txt",
    "b16_k_sweep.txt": f"{tag}_b16_k_sweep.txt", "f32_k_sweep.txt": f"{tag}_f32_k_sweep.txt", "ceilings.txt": f"{tag}_ceilings.txt",
}
for a, b in copies.items():
    shutil.copy(os.path.join(src, a), os.path.join(dst, b))
for pattern, name in (("stats/*/*kernel_stats.csv", f"{tag}_bench_kernel_stats.csv"), ("stats_single/*/*kernel_stats.csv", f"{tag}_bench_kernel_stats_single_stream.csv"),
                      ("iso/*/*kernel_stats.csv", f"{tag}_isolated_kernel_stats.csv"), ("gae_stats/*/*kernel_stats.csv", f"{tag}_gae_kernel_stats.csv"),
                      ("cfg5_stats/*/*kernel_stats.csv", f"{tag}_cfg5_bf16_kernel_stats_single_stream.csv")):
    shutil.copy(newest(pattern), os.path.join(dst, name))
for name in copies.values():
    if name.endswith(".json") and "bench" in name:  # one JSON line -> pretty-printed for reading
        p = os.path.join(dst, name)
        json.dump(json.loads(open(p).read().strip().splitlines()[-1]), open(p, "w"), indent=1)
print("profiles/%s_* written" % tag)
