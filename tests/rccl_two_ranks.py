"""Rank script of tests/test_gpu_two_ranks.py: the library's OWN communicator across processes on distinct GPUs
(lvbgpu_comm_unique_id on rank 0, the id handed over through a file, lvbgpu_comm_init, lvbgpu_allreduce_min with its
argmin rank, lvbgpu_allreduce_sum of per-shard partial lengths).  No torch.distributed: nothing but the C-ABI."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    rank, world, idfile = int(sys.argv[1]), int(sys.argv[2]), Path(sys.argv[3])
    from lvb_amd import api, host
    from tests import synth
    n, m = 24, 5000
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 9))
    m = len(rows[0])
    lo, hi = api.site_slice(m, rank, world)
    shard = api.FitchContext(text_rows=[r[lo:hi] for r in rows], device=rank)      # this rank's columns
    whole = api.FitchContext(text_rows=rows, device=rank)
    tree = host.HostTree(n, seed=3)                                                # the same tree on every rank
    tree.upload(shard)
    tree.upload(whole)
    if rank == 0:
        uid = api.comm_unique_id()
        tmp = idfile.with_suffix(".tmp")
        tmp.write_bytes(uid)
        tmp.rename(idfile)
    else:
        deadline = time.time() + 120
        while not idfile.exists():
            if time.time() > deadline:
                raise SystemExit("rank 0 never published the communicator id")
            time.sleep(0.05)
        uid = idfile.read_bytes()
    shard.comm_init(world, rank, uid)
    # independent restarts: the best length and a rank that holds it
    mine = 1_000_000 - 1000 * ((rank + 1) % world)                                  # smallest on rank world - 2
    best, who = shard.allreduce_min(mine)
    # site shards: candidate lengths are sums over the ranks' columns
    cands = [tree.propose(1) for _ in range(32)]                                   # same seed -> same moves on every rank
    partial = shard.score_batch(cands)
    total = shard.allreduce_sum(partial)
    print(json.dumps({"rank": rank, "best": int(best), "who": int(who), "total": [int(x) for x in total],
                      "whole": [int(x) for x in whole.score_batch(cands)]}), flush=True)
    shard.close()
    whole.close()
    tree.close()


if __name__ == "__main__":
    main()
