"""bench.py --gpus N starts its N ranks itself (CPU tier: the rank flow over gloo, no GPU, nothing scored)."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_gpus_2_spawns_two_ranks_and_prints_one_line():
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-ranks", "--steps", "3"],
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout                      # rank 0's line and nothing else
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry"] is True and d["scaling"] == "weak"
    assert d["config"]["best_length"] == 1000              # the minimum sat on rank 1: the reduce crossed ranks
    assert d["config"]["seed_sum"] == 3001 + 3002          # every rank had its own restart seed
    assert "lvbgpu_allreduce_min" in d["config"]["parallelism"]
    # the timed region measures steady state: one untimed min-reduce (+ barrier) BEFORE t0, so that the communicator's
    # lazy first-collective set-up is not read as a scaling loss; the in-region reduce is timed on its own
    assert d["config"]["order"] == ["reduce:warmup", "barrier", "t0", "steps", "reduce:timed", "t1"]
    assert d["config"]["reduces"] == 2 and d["config"]["reduce_ms"] >= 0.0
    assert d["config"]["per_rank"] == [1.0, 2.0] and d["config"]["comm_size"] == 2      # every rank's own value is in the line


def test_one_rank_needs_no_reduce():
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--dry-ranks", "--steps", "3"],
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    assert d["n_gpus"] == 1 and d["config"]["order"] == ["barrier", "t0", "steps", "reduce:timed", "t1"]
    assert d["config"]["reduces"] == 0 and d["config"]["reduce_ms"] == 0.0


def test_a_failing_rank_fails_the_run():
    # --gpus 2 against a WORLD_SIZE of 3 in the children cannot happen; a rank that exits non-zero can:
    # ask for the dry flow with a steps value argparse rejects in every child
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-ranks", "--move", "spr", "--seed", "x"],
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode != 0


def test_under_a_launcher_the_world_size_must_match():
    env = dict(_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29613")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-ranks"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr
