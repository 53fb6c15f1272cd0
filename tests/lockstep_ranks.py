"""Rank script of tests/test_lockstep_ranks_cpu.py: lvbhost_anneal_chains in lockstep mode (sync_every > 0) as one of
several processes, on the scorer's test double whose lvbgpu_allreduce_min meets the other ranks in a directory
(LVBGPU_DOUBLE_COMM_DIR / _RANK / _WORLD: set by the test).  Usage: lockstep_ranks.py rank chains max_proposals"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    rank, R, max_proposals = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    from lvb_amd import host
    from oracle import binding
    from tests import synth
    from tests.cpu_double import build
    binding.load_oracle()
    lib, new_ctx, free_ctx = build.load()
    n, m = 16, 400
    rows, min_len = host.prepare_alignment(synth.treelike_rows(n, m, 29), lib)      # the same alignment on every rank
    ctx = new_ctx(rows)
    trees = [host.HostTree(n, seed=100 * rank + c + 1, lib=lib) for c in range(R)]
    params = []
    for c in range(R):
        p = host.anneal_defaults(lib)
        p.seed = 500 * rank + c + 1
        p.algorithm = 1
        p.batch = 32
        p.t0 = 0.002                             # given: the run starts annealing at once
        p.min_len_tree = min_len
        p.max_proposals = max_proposals          # a rank with a small cap has nothing left to do long before the others
        p.max_device_steps = 400
        p.sync_every = 37
        p.log_cap = 8
        params.append(p)
    res, _ = host.anneal_chains(ctx, trees, params, lib=lib)
    print(json.dumps({"rank": rank, "best": [r["best_length"] for r in res], "global": [r["global_best_length"] for r in res],
                      "steps": [r["device_steps"] for r in res], "consumed": [r["consumed"] for r in res]}), flush=True)
    for t in trees:
        t.close()
    free_ctx(ctx)


if __name__ == "__main__":
    main()
