"""Rank script of tests/test_launch_gloo.py::test_site_axis_shards_sum_over_ranks: the site-axis sharding of
SURVEY.md 8(e) rehearsed on CPU - every rank scores the same trees on ITS column slice with the CPU oracle, the
partial lengths are summed over the ranks (gloo stands in for the RCCL sum of lvbgpu_allreduce_sum)."""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import torch
    from lvb_amd import api
    from lvb_amd.launch import Ranks
    from oracle import binding as ob
    from tests import helpers, synth
    r = Ranks(backend="gloo")
    n, m = 20, 7000
    rows = ob.cut_constant_columns(synth.treelike_rows(n, m, 5))
    m = len(rows[0])
    lo, hi = api.site_slice(m, r.rank, r.world)
    mine = ob.encode_rows([row[lo:hi] for row in rows])
    full = ob.encode_rows(rows)
    rng = np.random.default_rng(11)            # the same trees on every rank
    partial, whole = [], []
    for _ in range(6):
        # a random tree: leaves sprout on random edges
        left, right = np.full(2 * n - 3, -1, np.int64), np.full(2 * n - 3, -1, np.int64)
        left[0], right[0] = 1, 2
        parent = {1: 0, 2: 0}
        nxt = n
        for leaf in range(3, n):
            x = int(rng.choice(list(parent)))
            p = parent[x]
            if left[p] == x:
                left[p] = nxt
            else:
                right[p] = nxt
            parent[nxt] = p
            left[nxt], right[nxt] = x, leaf
            parent[x] = parent[leaf] = nxt
            nxt += 1
        par = helpers.parents_of(left, right)
        for enc, out in ((mine, partial), (full, whole)):
            t = ob.OracleTree(n, enc.shape[1], enc)
            t.set_topology(par, left, right, 0)
            out.append(t.getplen() if enc.shape[1] else 0)
    tens = torch.tensor(partial, dtype=torch.int64)
    r.dist.all_reduce(tens, op=r.dist.ReduceOp.SUM)
    print(json.dumps({"rank": r.rank, "slice": [lo, hi], "sum": tens.tolist(), "whole": whole}), flush=True)
    r.close()


if __name__ == "__main__":
    main()
