"""The library's own RCCL communicator across processes on DISTINCT GPUs (api_comm.cpp).  Needs two
visible devices: the pool's test boxes have one, where this file skips - it is for multi-GPU hosts (the driver's
8-GPU node runs the same calls through bench.py --gpus N)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_min_and_sum_over_two_gpus(tmp_path):
    from lvb_amd import api
    if api.device_count() < 2:
        pytest.skip("one GPU visible: RCCL refuses two ranks on one device")
    world = 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "rccl_two_ranks.py"), str(r), str(world),
                               str(tmp_path / "rccl.id")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in range(world)]
    outs = []
    try:
        for p in procs:
            out, err = p.communicate(timeout=600)
            assert p.returncode == 0, err[-2000:]
            outs.append(json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for o in outs:
        assert o["best"] == 1_000_000 - 1000 * (world - 1) and o["who"] == world - 2
        assert o["total"] == o["whole"]                     # partial lengths of the column shards add up
    assert outs[0]["total"] == outs[1]["total"]


def test_rank_script_alone_on_one_gpu(tmp_path):
    """The same script as a world of one (keeps it from rotting on one-GPU boxes)."""
    p = subprocess.run([sys.executable, str(ROOT / "tests" / "rccl_two_ranks.py"), "0", "1", str(tmp_path / "rccl.id")],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0, p.stderr[-2000:]
    o = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert o["best"] == 1_000_000 and o["who"] == 0 and o["total"] == o["whole"]
