#!/usr/bin/env python3
"""Randomised run of the multi-chain step on the GPU box: many small shapes (every taxon count from 5 to 140; site counts
around the word, tile and multi-tile boundaries), 1-5 chains, whole annealing steps (lvbgpu_chains_step_*: the library
decides and commits) in the two batch slots in turn, re-roots in between - so that commit walk, table rebuilds, re-roots
and the next generator go out as ONE post launch - and after every few steps each
chain against the C ORACLE: full evaluation of the topology the library reports == the resident length, per-node changes
and node sets; and the next device-drawn neighbourhood replayed on the host generators scores what the oracle scores.
Prints one line per shape only on failure, and a summary.

    gpurun -- python tests/manual/gpu_chains_fuzz.py [--seconds 200]
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
    ap.add_argument("--seconds", type=float, default=200.0)
    a = ap.parse_args()
    from lvb_amd import api, host
    from oracle import binding as ob
    from tests import synth
    from tests.helpers import apply_edits, parents_of
    rng = np.random.default_rng(4242)
    sites = [1, 2, 15, 16, 17, 31, 33, 100, 2047, 2048, 2049, 4096, 4100, 9000]
    t_end = time.perf_counter() + a.seconds
    shapes = checks = steps = posts = fused = 0
    k = 0
    bad = []
    while time.perf_counter() < t_end:
        n = 5 + k % 136
        m = sites[(k // 3) % len(sites)] if k % 3 else int(rng.integers(1, 6000))
        k += 1
        try:
            rows, min_len = host.prepare_alignment(synth.treelike_rows(n, m, 20_000 + k))
        except ValueError:
            continue  # every column constant
        enc = ob.encode_rows(rows)
        R = int(rng.integers(1, 6))
        ctx = api.FitchContext(text_rows=rows)
        ctx.set_chains(R)
        cur, roots = [], []
        for c in range(R):
            ctx.select_chain(c)
            cur.append(host.HostTree(n, seed=31 * k + c).upload(ctx))
            roots.append(ctx.topology()[3])

        def check(tag):
            nonlocal checks
            ok = True
            for c in range(R):
                ctx.select_chain(c)
                _, l, r, root = ctx.topology()
                l64, r64 = l.astype(np.int64), r.astype(np.int64)
                t = ob.OracleTree(n, enc.shape[1], enc)
                t.set_topology(parents_of(l64, r64), l64, r64, root)
                ok &= ctx.current_length() == t.getplen() == cur[c] and root == roots[c]
                ok &= bool(np.array_equal(ctx.changes()[n:], t.changes()[n:])) and bool(np.array_equal(ctx.all_sets(), t.all_sets()))
                # the generator's tables have followed: a fresh neighbourhood, replayed on the host, scores what the oracle scores
                lens = ctx.propose_score(5, -1, 7 * k + c)
                for b in range(5):
                    if lens[b] == np.iinfo(np.int64).max:
                        continue
                    edits, _ = ctx.proposal_edits(b)
                    nl, nr = apply_edits(l, r, edits)
                    cand = ob.OracleTree(n, enc.shape[1], enc)
                    cand.set_topology(parents_of(nl, nr), nl, nr, root)
                    ok &= int(lens[b]) == cand.getplen()
                    checks += 1
            if not ok:
                bad.append((n, m, R, tag))
            return ok

        ok = True
        for step in range(12):
            active = sorted(rng.choice(R, size=int(rng.integers(1, R + 1)), replace=False).tolist())
            # small draws (drawn by the rebuilding workgroup itself) and, now and then, one its generator workgroups wait for
            draws = [(c, int(rng.integers(1, 40)) if (step + c) % 4 else int(rng.integers(70, 200)), [1, 2, -1, 0][(step + c) % 4],
                      1000 * k + 10 * step + c) for c in active]
            rules = [(cur[c], [1e-9, 2e-5, 4e-4, 5e-2][(step + c) % 4], float(min_len), 31 * step + c) for c in active]
            lens, picks = ctx.chains_step(draws, rules, slot=step & 1)
            steps += 1
            for i, c in enumerate(active):
                if picks[i] >= 0:
                    cur[c] = int(lens[i][picks[i]])
            who = [c for i, c in enumerate(active) if (picks[i] < 0 or rng.random() < 0.2) and rng.random() < 0.5]
            if who:
                reqs = []
                for c in who:
                    roots[c] = int((roots[c] + 1 + rng.integers(0, n - 1)) % n)
                    reqs.append((c, roots[c]))
                ctx.chains_reroot(reqs)
            if step % 4 == 3:
                ok &= check(step)
        p, g = ctx.post_launches()
        posts += p
        fused += g
        ctx.close()
        shapes += 1
    print(f"{shapes} shapes, {steps} steps, {posts} post launches ({fused} with the next generator), {checks} oracle-checked candidates, "
          f"{len(bad)} failures")
    for b in bad[:20]:
        print("FAILED (n, m, chains, step):", b)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
