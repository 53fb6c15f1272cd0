#!/usr/bin/env python3
"""Parity beyond 4 GiB: a tree block whose byte offsets do not fit 32 bits (default 1200 taxa x 4 194 304
sites: 2397 rows x 2 MiB = 5.0 GB resident), checked against the C oracle (test infrastructure) on the
GPU box: full evaluation, a batch of SPR/TBR candidates scored incrementally, one commit, node sets of
the highest rows.  One JSON line.  Not a pytest: it needs ~12 GB of host memory and a few minutes.

    gpurun -- python tests/manual/big_parity.py
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--taxa", type=int, default=1200)
    ap.add_argument("--sites", type=int, default=4194304)
    ap.add_argument("--cands", type=int, default=48)
    a = ap.parse_args()
    from lvb_amd import api, host
    from oracle import binding as ob
    from tests.helpers import apply_edits, parents_of
    n, m = a.taxa, a.sites
    nwords = (m + 15) // 16
    t0 = time.perf_counter()
    rng = np.random.default_rng(7)
    enc = np.zeros((n, nwords), dtype=np.uint64)
    base = rng.integers(0, 4, size=nwords * 16, dtype=np.uint8)          # a common ancestor ...
    shifts = (np.arange(16, dtype=np.uint64) * np.uint64(4))
    for i in range(n):                                                   # ... and 8 % substitutions per taxon
        row = base.copy()
        hit = rng.random(row.size) < 0.08
        row[hit] = rng.integers(0, 4, size=int(hit.sum()), dtype=np.uint8)
        nib = (np.uint64(1) << row.astype(np.uint64)).reshape(nwords, 16)
        if m % 16:
            nib.reshape(-1)[m:] = np.uint64(15)                            # padding sites are N
        enc[i] = (nib << shifts).sum(axis=1, dtype=np.uint64)
        if i % 100 == 99:
            print(f"[big_parity] row {i + 1}/{n}", file=sys.stderr, flush=True)
    t_gen = time.perf_counter() - t0
    print(f"[big_parity] generated {n} x {m} in {t_gen:.0f} s", file=sys.stderr, flush=True)
    tree = host.HostTree(n, seed=3)
    _, left, right = tree.arrays()
    l64, r64 = left.astype(np.int64), right.astype(np.int64)
    out = {"taxa": n, "sites": m, "nwords": nwords, "rows": 2 * n - 3,
           "tree_block_bytes": int((2 * n - 3) * ((nwords + 127) // 128 * 128) * 8), "gen_s": round(t_gen, 1)}
    # oracle (CPU)
    t0 = time.perf_counter()
    ot = ob.OracleTree(n, nwords, enc)
    ot.set_topology(parents_of(l64, r64), l64, r64, tree.root)
    want_full = ot.getplen()
    out["oracle_full_s"] = round(time.perf_counter() - t0, 2)
    print(f"[big_parity] oracle full evaluation {out['oracle_full_s']} s", file=sys.stderr, flush=True)
    # device
    t0 = time.perf_counter()
    ctx = api.FitchContext(enc)
    got_full = tree.upload(ctx)
    out["device_create_and_full_s"] = round(time.perf_counter() - t0, 2)
    out["full_length"] = int(got_full)
    out["full_matches"] = bool(got_full == want_full)
    print(f"[big_parity] device full evaluation: {got_full} (oracle {want_full})", file=sys.stderr, flush=True)
    cands = [tree.propose(1 + (k & 1)) for k in range(a.cands)]
    t0 = time.perf_counter()
    got = ctx.score_batch(cands)
    out["score_batch_s"] = round(time.perf_counter() - t0, 3)
    prop = ob.OracleTree(n, nwords)
    ok = True
    t0 = time.perf_counter()
    for e, g in zip(cands[:12], got[:12]):   # the oracle needs a 5 GB treecopy per candidate: a dozen will do
        nl, nr = apply_edits(left, right, e)
        prop.copy_from(ot)
        prop.set_topology(parents_of(nl.astype(np.int64), nr.astype(np.int64)), nl.astype(np.int64), nr.astype(np.int64),
                          tree.root)
        prog = tree.program(mode=0, edits=e)
        prop.mark_dirty([d for d in prog["dsts"] if d >= 0])
        ok &= bool(prop.getplen() == g)
        print("[big_parity] candidate checked", file=sys.stderr, flush=True)
    out["oracle_candidates_s"] = round(time.perf_counter() - t0, 1)
    out["candidates_match"] = ok
    # commit the first candidate, compare the length and the node sets of the last (highest-address) rows
    length = ctx.commit(cands[0])
    nl, nr = apply_edits(left, right, cands[0])
    ot.set_topology(parents_of(nl.astype(np.int64), nr.astype(np.int64)), nl.astype(np.int64), nr.astype(np.int64), tree.root)
    ot.mark_dirty([d for d in tree.program(mode=0, edits=cands[0])["dsts"] if d >= 0])
    out["commit_matches"] = bool(ot.getplen() == length == got[0])
    sets_ok = True
    for node in (2 * n - 4, 2 * n - 5, n, n + 1):
        sets_ok &= bool(np.array_equal(ctx.sets(node), ot.sets(node)))
    out["node_sets_match"] = sets_ok
    out["ok"] = bool(out["full_matches"] and ok and out["commit_matches"] and sets_ok)
    print(json.dumps(out))
    return 0 if out["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
