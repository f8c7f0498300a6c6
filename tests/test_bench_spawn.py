"""bench.py --gpus N with no torchrun environment must start its N ranks itself (fresh children of torch.distributed.run,
parent GPU-free) and relay rank 0's JSON line.  RLPPO_BENCH_DRYRUN=2 stops every rank after the rendezvous and one CPU
all-reduce, so the launch path is checked without a GPU (the real runs: one rank per GPU over RCCL, driver-launched)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env):
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, timeout=300)


def test_bare_invocation_spawns_its_own_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"RLPPO_BENCH_DRYRUN": "2"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                      # exactly ONE JSON line on stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["hip_initialised"] is False               # the dry-run ranks made no GPU call either
    assert "starting 2 ranks" in r.stderr and "parent initialised HIP: False" in r.stderr


def test_three_ranks_and_argument_relay():
    r = _run(["--gpus", "3", "--steps", "7", "--warmup", "2", "--no-extras"], {"RLPPO_BENCH_DRYRUN": "2"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 3 and line["steps"] == 7


def test_eight_ranks_the_drivers_top_configuration():
    """N = 8 (BASELINE configs[3], the last point of the driver's scaling run) through the same launch path: 8 fresh children of
    torch.distributed.run, rendezvous on 127.0.0.1, one collective spanning all 8, ONE JSON line with n_gpus = 8."""
    r = _run(["--gpus", "8", "--steps", "2", "--warmup", "1"], {"RLPPO_BENCH_DRYRUN": "2", "OMP_NUM_THREADS": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["hip_initialised"] is False


@pytest.mark.gpu
def test_two_rank_dry_run_on_the_one_gpu_walks_the_whole_multi_rank_path():
    """[r5] What first executes on the driver's 8-GPU node, as far as a one-GPU box can take it: `bench.py --gpus 2` through its
    own spawn path with both ranks on cuda:0 and gloo as the collective (RLPPO_BENCH_DRYRUN=1; RCCL refuses two ranks on one GPU):
    rendezvous + init watchdog, the warm all-reduce probe, PPOLearner.learn with world = 2 (slice dealing, the gradient exchange
    before clipping, the statistics exchange with the give-up word), max-over-ranks timing and ONE JSON line whose fields say
    what ran.  The number is meaningless and tagged so."""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "1", "--no-extras", "--epochs", "2"], {"RLPPO_BENCH_DRYRUN": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 1 and line["metric"] == "ppo_update_samples_per_sec" and line["value"] > 0
    assert "DRY RUN" in line["data"] and line["config"]["parallelism"].startswith("dp2: 4 minibatch slice(s) per rank")
    assert line["allreduce_us"] > 0 and line["scaling"] == "strong"
    pr = line["per_rank_ms_per_step"]
    assert len(pr["all"]) == 2 and pr["min"] <= pr["max"] and abs(pr["max"] - line["ms_per_step"]) <= 0.05 * line["ms_per_step"] + 1.0
    assert "exiting 4" not in r.stderr                               # the init watchdog did not trip
    assert line["config"]["last_report"]["Cumulative Model Updates"] == 4   # (warm-up + 1 step) x 2 epochs x 1 batch


@pytest.mark.gpu
def test_process_collect_leg_runs_and_counts_what_it_says():
    """[r5] bench.py's process_collect leg at a small size (3 worker processes, 3,000 timesteps): it returns the keys the line
    carries, collected at least what was asked, counted get_action calls whose sizes add up to the timesteps, and left no worker
    behind.  (The worker processes import the main module: the test runs the leg in a child interpreter, as bench.py does.)"""
    code = ("import json, bench\n"
            "r = bench.process_collect_leg(3, 3000, limit_s=90)\n"
            "print(json.dumps(r))\n")
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert "error" not in out, out
    assert out["n_proc"] == 3 and out["timesteps"] >= 3000 and out["steps_per_s"] > 1000
    assert out["get_action_calls"] > 0 and abs(out["mean_obs_per_call"] * out["get_action_calls"] - out["timesteps"]) <= 6 * 8
    assert 0.0 < out["frac_of_wall_in_get_action"] < 1.0 and out["us_per_get_action_median"] > 5
    # [r6] the transport's counters cover every call of the timed collection and the 1 % spot check against the general path is clean
    tr = out["transport"]
    assert tr["calls"] >= out["get_action_calls"] and tr["polled"] + tr["poll_timeouts"] >= tr["calls"]
    assert out["spot_checked"] >= out["get_action_calls"] // 100 and out["spot_check_mismatches"] == 0
    # [r6] the leg also runs the rest of the reference's iteration on what it collected, and says which collection loop served it
    assert out["collector"].startswith("C++") and out["add_new_experience_ms"] > 0 and out["learn_ms"] > 0 and out["iteration_steps_per_s"] > 0
