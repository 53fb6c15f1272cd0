#!/usr/bin/env python3
"""Randomised parity run on the GPU box: many small shapes (every taxon count from 5 to 140, so whole-tree
programs straddle the 64-token chunk boundary in every way; site counts around the word, tile and
multi-tile boundaries), each checked against the C oracle: full evaluation, node sets, a mixed batch of
NNI/SPR/TBR candidates scored from edits and from move parameters, drawn-on-device candidates replayed,
commits, re-roots.  Prints one line per shape only on failure, and a summary.

    gpurun -- python tests/manual/gpu_fuzz.py [--seconds 240]
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
    ap.add_argument("--seconds", type=float, default=240.0)
    a = ap.parse_args()
    from lvb_amd import api, host
    from oracle import binding as ob
    from tests import synth
    from tests.helpers import apply_edits, parents_of
    rng = np.random.default_rng(2025)
    sites = [1, 2, 15, 16, 17, 31, 33, 100, 2047, 2048, 2049, 4096, 4100, 9000]
    t_end = time.perf_counter() + a.seconds
    shapes = checks = 0
    n_list = list(range(5, 141))
    k = 0
    while time.perf_counter() < t_end:
        n = n_list[k % len(n_list)]
        m = sites[(k // 3) % len(sites)] if k % 3 else int(rng.integers(1, 6000))
        k += 1
        rows = synth.treelike_rows(n, m, 10_000 + k)
        try:
            rows, _ = host.prepare_alignment(rows)
        except ValueError:
            continue  # every column constant
        enc = ob.encode_rows(rows)
        ctx = api.FitchContext(text_rows=rows)
        tree = host.HostTree(n, seed=k)
        uni = host.RefRng(k % 900000)
        length = tree.upload(ctx)
        _, left, right = tree.arrays()
        ot = ob.OracleTree(n, enc.shape[1], enc)
        ot.set_topology(parents_of(left.astype(np.int64), right.astype(np.int64)), left.astype(np.int64),
                        right.astype(np.int64), tree.root)
        ok = length == ot.getplen() and np.array_equal(ctx.all_sets(), ot.all_sets())
        for round_ in range(3):
            _, left, right = tree.arrays()
            B = int(rng.integers(1, 80))
            moves = [tree.ref_draw_move(uni, int(rng.integers(0, 3))) for _ in range(B)]
            cands = [tree.move_edits(mv) for mv in moves]
            got = ctx.score_batch(cands)
            ok &= bool(np.array_equal(got, ctx.score_moves(moves)))
            prop = ob.OracleTree(n, enc.shape[1])
            for e, g in list(zip(cands, got))[:6]:
                nl, nr = apply_edits(left, right, e)
                prop.copy_from(ot)
                prop.set_topology(parents_of(nl.astype(np.int64), nr.astype(np.int64)), nl.astype(np.int64),
                                  nr.astype(np.int64), tree.root)
                prop.mark_dirty([d for d in tree.program(mode=0, edits=e)["dsts"] if d >= 0])
                ok &= bool(prop.getplen() == g)
                checks += 1
            # candidates drawn on the device replay on the host
            dev = ctx.propose_score(16, -1, 77 + k)
            for b in (0, 5, 15):
                e, _info = ctx.proposal_edits(b)
                if dev[b] < 2**62:
                    ok &= bool(ctx.score_batch([e])[0] == dev[b])
            # accept one, sometimes re-root, and compare the resident state with the oracle
            pick = int(rng.integers(0, B))
            ok &= bool(ctx.commit(cands[pick]) == got[pick])
            tree.apply(cands[pick])
            if round_ == 1:
                e, nr_ = tree.ref_arbreroot(uni)
                ctx.commit(e, nr_)
                tree.apply(e, nr_)
            _, left, right = tree.arrays()
            ot.set_topology(parents_of(left.astype(np.int64), right.astype(np.int64)), left.astype(np.int64),
                            right.astype(np.int64), tree.root)
            ot.mark_all_dirty()
            ok &= bool(ot.getplen() == ctx.current_length()) and bool(np.array_equal(ctx.all_sets(), ot.all_sets()))
            ok &= bool(np.array_equal(ctx.changes()[n:], ot.changes()[n:]))
        ctx.close()
        shapes += 1
        if not ok:
            print(f"MISMATCH at n={n} m={m} (shape #{k})", flush=True)
            return 1
        if shapes % 50 == 0:
            print(f"[gpu_fuzz] {shapes} shapes, {checks} candidate checks", file=sys.stderr, flush=True)
    print(f"gpu_fuzz ok: {shapes} shapes, {checks} oracle-checked candidates, no mismatch")
    return 0


if __name__ == "__main__":
    sys.exit(main())
