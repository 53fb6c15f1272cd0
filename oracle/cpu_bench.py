#!/usr/bin/env python3
"""Time LVB's CPU path (the compiled reference in oracle/_ref) on one core; bench.py runs one copy of
this per host core for the all-cores figure.  TEST/BENCH INFRASTRUCTURE ONLY.

  python -m oracle.cpu_bench --taxa 500 --sites 50000 --seed 3 --kind 1 --seconds 8 [--tree tree.npz]

With --tree (parent/left/right/root arrays) every process scores random neighbours of THAT tree and accepts
nothing, so the dirty-path lengths are those of the GPU leg that scored the same tree; without it the chain
starts from the reference's own random tree and accepts every 4th proposal.
"""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--taxa", type=int, required=True)
    ap.add_argument("--sites", type=int, required=True)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--dist", choices=["tree", "uniform"], default="tree")
    ap.add_argument("--kind", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--chain", type=int, default=0)
    ap.add_argument("--tree", default=None)
    a = ap.parse_args()
    import numpy as np

    from oracle import binding as ob
    from tests.synth import treelike_rows, uniform_rows
    if ob.load_ref() is None:
        print(json.dumps({"error": "no reference"}))
        return
    rows = treelike_rows(a.taxa, a.sites, a.seed) if a.dist == "tree" else uniform_rows(a.taxa, a.sites, a.seed)
    rr = ob.RefRun(rows=rows, seed=12345 + a.chain, nproc=1)
    accept_every = 4
    if a.tree:
        t = np.load(a.tree)
        rr.set_topology(t["parent"], t["left"], t["right"], int(t["root"]))
        accept_every = 0
    rr.getplen(0)
    tg, tm, _, _ = rr.time_proposals(a.kind, 20, accept_every)
    reps = int(max(50, min(50000, a.seconds / max((tg + tm) / 20, 1e-6))))
    tg, tm, cs, ds = rr.time_proposals(a.kind, reps, accept_every)
    print(json.dumps({"reps": reps, "t_getplen": tg, "t_mutate": tm, "dirty": ds / reps, "checksum": cs}))
    rr.close()


if __name__ == "__main__":
    main()
