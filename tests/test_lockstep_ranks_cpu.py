"""Lockstep annealing over several ranks (lvbhost_anneal_chains with sync_every > 0: every rank runs exactly
max_device_steps device steps and min-reduces the best length every sync_every of them), rehearsed on the CPU: three
processes on the scorer's test double, whose lvbgpu_allreduce_min meets the other ranks in a directory.  Ranks with
different numbers of chains, one of which runs out of proposals long before the others: the collectives must still pair
up (a rank that stopped calling would leave the others waiting), and every rank must end with the same global best."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_three_ranks_in_lockstep_agree_on_the_global_best(tmp_path):
    from oracle import binding
    binding.load_oracle()
    from tests.cpu_double import build
    build.build()                                   # once, before the ranks race to do it
    world = 3
    shape = [(3, 100000), (1, 150), (2, 100000)]     # (chains, proposal cap) per rank
    procs = []
    for r, (chains, cap) in enumerate(shape):
        env = dict(os.environ, LVBGPU_DOUBLE_COMM_DIR=str(tmp_path), LVBGPU_DOUBLE_RANK=str(r), LVBGPU_DOUBLE_WORLD=str(world))
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "tests" / "lockstep_ranks.py"), str(r), str(chains), str(cap)],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = []
    try:
        for p in procs:
            out, err = p.communicate(timeout=300)
            assert p.returncode == 0, err[-2000:]
            outs.append(json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    best_of_all = min(b for o in outs for b in o["best"])
    for o in outs:
        assert set(o["global"]) == {best_of_all}, outs        # every chain of every rank reports the same global best
    assert outs[1]["consumed"] == [150]                       # the short rank stopped proposing ...
    assert max(outs[0]["consumed"]) > 150                     # ... while the others went on
    # 400 device steps, a reduce every 37th and at the end: 11 calls per rank, all of them paired
    assert sorted(p.name for p in tmp_path.iterdir() if not p.name.startswith("tmp_")) == sorted(
        f"{k}_{r}" for k in range(11) for r in range(world))
