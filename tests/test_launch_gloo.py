"""The N > 1 plumbing on CPU: two processes, gloo backend, rendezvous on 127.0.0.1."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_share_id_reduce_and_get_distinct_seeds():
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PYTHONPATH=str(ROOT))
        procs.append(subprocess.Popen([sys.executable, "-m", "lvb_amd.launch"], cwd=ROOT, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=240)
        assert p.returncode == 0, err[-2000:]
        outs.append(json.loads(out.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert [d["world"] for d in outs] == [2, 2]
    assert outs[0]["seed"] != outs[1]["seed"]                       # independent restarts
    assert all(d["token"] == "id-from-rank-0" for d in outs)        # RCCL id travels from rank 0
    assert all(d["max"] == 2.0 for d in outs)                       # max-over-ranks timing
    assert all(d["sum"] == 21 for d in outs)


def test_site_axis_shards_sum_over_ranks():
    """The other sharding (site axis, lvbgpu_allreduce_sum): two ranks, each with its column slice, partial lengths
    summed over gloo equal the whole alignment's lengths."""
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PYTHONPATH=str(ROOT))
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "tests" / "gloo_site_shards.py")], cwd=ROOT, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=240)
        assert p.returncode == 0, err[-2000:]
        outs.append(json.loads(out.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert outs[0]["slice"][1] == outs[1]["slice"][0] and outs[0]["slice"][0] == 0      # the slices tile the alignment
    for d in outs:
        assert d["sum"] == d["whole"] and min(d["whole"]) > 0
