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


def test_eight_ranks_share_the_hosts_cores():
    """The driver's scaling run is --gpus 8 on one node: rehearsed over gloo.  Every rank keeps to ITS share of the cores
    the job may run on (affinity mask / LOCAL_WORLD_SIZE, set before the scoring library and its thread pool are loaded)
    and sizes the pool to it: eight ranks with sixteen free-running workers each is how a >= 6x target is lost on the host."""
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--dry-ranks", "--steps", "2"],
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    c = d["config"]
    assert d["n_gpus"] == 8 and c["comm_size"] == 8 and c["own_comm"] is True
    assert c["best_length"] == 1000 and c["per_rank"] == [float(r + 1) for r in range(8)]
    cores = len(os.sched_getaffinity(0))
    assert sum(c["cores_of_ranks"]) <= cores or cores < 8          # the shares do not overlap ...
    assert all(v >= 1 for v in c["cores_of_ranks"])
    assert c["pinned_to_share_of_cores"] == (cores >= 8 and cores // 8 < cores)
    assert 2 <= c["host_threads_per_rank"] <= max(2, min(16, int(max(c["cores_of_ranks"]))))


def test_a_rank_whose_communicator_never_comes_up_does_not_hang_the_run():
    """lvbgpu_comm_init is collective and has no deadline of its own (ncclCommInitRank): bench.py waits for it under one.
    Rehearsed: rank 1's set-up never returns - every rank agrees on the fallback, the line is printed, the run ends with 0."""
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "3", "--dry-ranks", "--steps", "2", "--dry-stuck-rank", "1"],
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    assert d["n_gpus"] == 3 and d["config"]["own_comm"] is False and d["config"]["best_length"] == 1000
