#!/usr/bin/env python3
"""Soak of the two searches on the GPU box: every move schedule (-a 0/1/2 and the single-kind ones) x every
proposal mode (host, device, automatic) x a few seeds and shapes must run to completion with sane lengths, and
the batched search's final tree must score (full evaluation on a fresh context) what the search says it does.
Found nothing since the stale-length-slot fix; kept because that bug only showed in such runs.

    gpurun -- python tests/manual/search_soak.py [--seconds 150]
"""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=150.0)
    a = ap.parse_args()
    from lvb_amd import api, host
    from tests import synth
    t_end = time.time() + a.seconds
    runs = fails = 0
    for n, m in ((500, 50000), (200, 20000), (60, 3000), (33, 2049)):
        rows, minlen = host.prepare_alignment(synth.treelike_rows(n, m, 3))
        ctx = api.FitchContext(text_rows=rows)
        check = api.FitchContext(text_rows=rows)
        for seed in (5, 6, 7):
            for alg in (1, 0, 2, 10, 11, 12):
                for devp in (2, 0, 1):
                    if time.time() > t_end:
                        break
                    tree = host.HostTree(n, seed=seed)
                    start = tree.upload(ctx)
                    p = host.anneal_defaults()
                    p.seed = seed
                    p.min_len_tree = minlen
                    p.algorithm = alg
                    p.device_proposals = devp
                    p.max_seconds = 3.0
                    runs += 1
                    try:
                        res, _ = host.anneal(ctx, tree, p)
                        _, left, right = tree.arrays()
                        full = check.set_tree(left, right, tree.root)
                        ok = 0 < res["best_length"] <= start and res["final_length"] == full == ctx.current_length()
                    except api.LvbGpuError as exc:
                        ok = False
                        res = {"error": str(exc)}
                    if not ok:
                        fails += 1
                        print(f"FAIL {n}x{m} seed {seed} -a {alg} proposals {devp}: {res}", flush=True)
                    tree.close()
            # the reference's trajectory as well (no reference binary here: completion and a consistent final tree)
            for alg in (0, 1, 2):
                if time.time() > t_end:
                    break
                p = host.refsearch_defaults()
                p.seed = seed
                p.algorithm = alg
                p.min_len_tree = minlen
                runs += 1
                try:
                    res, tree = host.reference_search(ctx.h, p)
                    _, left, right = tree.arrays()
                    full = check.set_tree(left, right, tree.root)
                    ok = res["final_length"] == full and 0 < res["best_length"] <= res["start_length"]
                    tree.close()
                except api.LvbGpuError as exc:
                    ok = False
                    res = {"error": str(exc)}
                if not ok:
                    fails += 1
                    print(f"FAIL exact {n}x{m} seed {seed} -a {alg}: {res}", flush=True)
        print(f"[soak] {n}x{m} done: {runs} runs so far, {fails} failures", flush=True)
        ctx.close()
        check.close()
    print(f"search_soak {'ok' if fails == 0 else 'FAILED'}: {runs} runs, {fails} failures")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
