"""bench.py --gpus N with no torchrun environment must start its N ranks itself (fresh children of torch.distributed.run,
parent GPU-free) and relay rank 0's JSON line.  RLPPO_BENCH_DRYRUN=2 stops every rank after the rendezvous and one CPU
all-reduce, so the launch path is checked without a GPU (the real runs: one rank per GPU over RCCL, driver-launched)."""
import json
import os
import subprocess
import sys

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
