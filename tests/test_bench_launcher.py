"""`python bench.py --gpus N` from a bare shell must start its own ranks: the launcher is rehearsed
here on the CPU with 2 gloo ranks and a tiny model (`--selftest`); what is under test is the rank plumbing - environment,
rank 0's single JSON line relayed by the parent, non-zero exit when a rank fails - not a measurement."""

import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GLR_FORCE_DIST")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest", "--steps", "3",
                           "--warmup", "1", "--global-batch", "8", *extra], env=env, capture_output=True, text=True,
                          timeout=300)


def test_launcher_starts_two_ranks_and_relays_one_json_line():
    r = _run()
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["world_size_seen"] == 2
    assert rec["config"]["per_gpu_batch"] == 4 and rec["steps"] == 3 and rec["warmup"] == 1
    assert rec["value"] > 0 and rec["ms_per_step"] > 0


def test_launcher_fails_when_a_rank_fails():
    r = _run("--selftest-fail-rank", "1")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_runs_as_the_rank_an_external_launcher_gives_it():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest"], env=env,
                       capture_output=True, text=True, timeout=300)
    # WORLD_SIZE is set, so bench.py runs as a rank: the selftest then simply runs with the world it was given
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1
