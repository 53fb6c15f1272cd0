#!/usr/bin/env python3
"""Soak of lvbhost_anneal_chains on the GPU box: shapes x move schedules x numbers of chains x run lengths
(lvbhost_anneal_params::run_levels: runs of accepted moves per step from cumulative host-made candidates) x seeds.
Every run must end with the resident tree of every chain equal to the host's mirror, its resident length equal to what
the chain believes AND to a full evaluation of that topology on a fresh context; runs with run_levels 1, 3 and 5 must
give the same results (the trajectory does not depend on the run length).

    gpurun -- python tests/manual/chains_soak.py [--seconds 600]
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=600.0)
    a = ap.parse_args()
    from lvb_amd import api, host
    from tests import synth
    shapes = [(12, 300), (40, 2000), (100, 6000), (200, 20000), (33, 2049)]
    algorithms = [0, 1, 2, 10, 11, 12]
    keys = ("start_length", "best_length", "final_length", "consumed", "accepted", "temperatures", "reroots", "topologies")
    t_end = time.perf_counter() + a.seconds
    runs = failures = 0
    k = 0
    last_print = time.perf_counter()
    while time.perf_counter() < t_end:
        n, m = shapes[k % len(shapes)]
        alg = algorithms[(k // len(shapes)) % len(algorithms)]
        R = (1, 2, 5)[(k // 7) % 3]
        seed0 = 100 + k
        k += 1
        rows, min_len = host.prepare_alignment(synth.treelike_rows(n, m, 500 + k % 17))
        per_levels = {}
        for levels in (0, 1, 3, 5):
            ctx = api.FitchContext(text_rows=rows)
            trees = [host.HostTree(n, seed=seed0 * 10 + c) for c in range(R)]
            params = []
            for c in range(R):
                p = host.anneal_defaults()
                p.seed = seed0 * 31 + c
                p.algorithm = alg
                p.batch = 256
                p.t0 = 0.0
                p.min_len_tree = min_len
                p.max_proposals = 4000 if n <= 100 else 2500
                p.log_cap = 16
                p.run_levels = levels
                params.append(p)
            try:
                res, _ = host.anneal_chains(ctx, trees, params)
                fresh = api.FitchContext(text_rows=rows)
                for c, t in enumerate(trees):
                    ctx.select_chain(c)
                    _, l, r = t.arrays()
                    _, ll, lr, lroot = ctx.topology()
                    assert np.array_equal(l, ll) and np.array_equal(r, lr) and t.root == lroot, "mirror"
                    assert ctx.current_length() == res[c]["final_length"], "resident length"
                    assert fresh.set_tree(l, r, t.root) == res[c]["final_length"], "full evaluation"
                    assert res[c]["consumed"] == params[c].max_proposals or res[c]["frozen"], "ran out early"
                    assert levels == 0 or res[c]["host_steps"] > 0, "no host-drawn step"
                fresh.close()
                per_levels[levels] = [{kk: r[kk] for kk in keys} for r in res]
            except Exception as e:  # noqa: BLE001 - a soak reports and goes on
                failures += 1
                print(f"[chains_soak] FAILED {n}x{m} alg {alg} R {R} levels {levels} seed {seed0}: {e!r}", flush=True)
            finally:
                for t in trees:
                    t.close()
                ctx.close()
            runs += 1
        if 1 in per_levels and (per_levels.get(3) != per_levels[1] or per_levels.get(5) != per_levels[1]):
            failures += 1
            print(f"[chains_soak] FAILED {n}x{m} alg {alg} R {R} seed {seed0}: run lengths disagree", flush=True)
        if time.perf_counter() - last_print > 30:
            last_print = time.perf_counter()
            print(f"[chains_soak] {runs} runs so far, {failures} failures", flush=True)
    print(f"chains_soak {'ok' if failures == 0 else 'FAILED'}: {runs} runs, {failures} failures", flush=True)
    return 0 if failures == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
