#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/vectors/ from the REAL reference.

Runs only where /root/reference exists (oracle/Makefile builds oracle/_ref/liblvbref.so from the
reference sources in place).  What is committed is DATA: encoded alignments, topologies, dirty
flags, and the lengths / per-node `changes` / node sets the reference's getplen produced.

  python tests/golden/gen_golden.py

One .npz per alignment.  Arrays (C = number of cases, nb = 2n-3):
  n, m, nwords, min_len_tree     dims after the reference's constant-column cut
  text  [n, m] uint8             alignment rows after the cut (small alignments only)
  synth [3] int64                (n, m, seed) of tests/synth.treelike_rows for regenerated ones
  enc   [n, nwords] uint64       DNAToBinary output (small alignments), enc_crc always
  kind  [C]   0 full(random tree) 1 NNI 2 SPR 3 TBR 4 re-root (arbreroot: everything dirty)
  base  [C]   case whose topology is the clean current tree this case was mutated from (-1: none)
  left, right [C, nb], root [C]  topology of the case
  dirty [C, nb] bool             sitestate[0]==0 flags right before getplen
  length [C], changes [C, nb]    what getplen returned / left in the tree block
  sets_crc [C]                   crc32 of all node sets after getplen; sets [C, nb, nwords] if small
  threads_length [C]             the same call through the OpenMP branch (nthreads > 1), or -1
"""
from __future__ import annotations

import sys
import zlib
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from oracle import binding as ob  # noqa: E402
from tests import synth  # noqa: E402

OUT = Path(__file__).resolve().parent / "vectors"
REFTESTS = Path(__file__).resolve().parent / "ref_tests"


def crc(a: np.ndarray) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def record(rr: ob.RefRun, which: int, kind: int, base: int, dirty, store_sets: bool, out: dict):
    length = rr.getplen(which)
    _, l, r, ch, _ = rr.tree(which)
    sets = rr.all_sets(which)
    out["kind"].append(kind)
    out["base"].append(base)
    out["left"].append(l.astype(np.int32))
    out["right"].append(r.astype(np.int32))
    out["root"].append(rr.root(which))
    out["dirty"].append(np.asarray(dirty, dtype=bool))
    out["length"].append(length)
    out["changes"].append(ch)
    out["sets_crc"].append(crc(sets))
    if store_sets:
        out["sets"].append(sets)
    return len(out["kind"]) - 1


def generate(name: str, rr: ob.RefRun, n_trees: int, n_moves: int, store_sets: bool, store_text: bool,
             synth_params=None, threads: int = 0):
    out = {k: [] for k in ["kind", "base", "left", "right", "root", "dirty", "length", "changes", "sets_crc",
                           "sets", "threads_length"]}
    nb = rr.nbranches
    all_dirty = np.zeros(nb, dtype=bool)
    all_dirty[rr.n:] = True
    for t in range(n_trees):
        if t:
            rr.random_tree()
        cur = record(rr, 0, 0, -1, all_dirty, store_sets, out)
        out["threads_length"].append(-1)
        for s in range(n_moves):
            kind = s % 3
            if rr.n < 5:
                break
            rr.mutate(kind)
            dirty = rr.tree(1)[4]
            idx = record(rr, 1, 1 + kind, cur, dirty, store_sets, out)
            out["threads_length"].append(-1)
            if s % 3 == 2:
                rr.swap()
                cur = idx
            if s % 7 == 6:
                rr.arbreroot()
                cur = record(rr, 0, 4, cur, all_dirty, store_sets, out)
                out["threads_length"].append(-1)
    if threads > 1:
        # the OpenMP site-slice branch must give the same answers (TreeEvaluation.c:64-181)
        rr.set_threads(threads)
        if rr.nthreads > 1:
            rr.random_tree()
            cur = record(rr, 0, 0, -1, all_dirty, store_sets, out)
            out["threads_length"].append(out["length"][-1])
            for s in range(6):
                rr.mutate(s % 3)
                dirty = rr.tree(1)[4]
                record(rr, 1, 1 + s % 3, cur, dirty, store_sets, out)
                out["threads_length"].append(out["length"][-1])
        rr.set_threads(1)

    enc = rr.enc()
    arrays = {
        "n": rr.n, "m": rr.m, "nwords": rr.nwords, "min_len_tree": rr.min_len, "original_m": rr.original_m,
        "nthreads_used": rr.nthreads if threads > 1 else 1,
        "enc_crc": crc(enc),
        "kind": np.array(out["kind"], dtype=np.int8), "base": np.array(out["base"], dtype=np.int32),
        "left": np.stack(out["left"]), "right": np.stack(out["right"]),
        "root": np.array(out["root"], dtype=np.int32), "dirty": np.stack(out["dirty"]),
        "length": np.array(out["length"], dtype=np.int64), "changes": np.stack(out["changes"]),
        "sets_crc": np.array(out["sets_crc"], dtype=np.uint32),
        "threads_length": np.array(out["threads_length"], dtype=np.int64),
    }
    if store_text:
        arrays["text"] = np.stack([np.frombuffer(r, dtype=np.uint8) for r in rr.rows()])
        arrays["enc"] = enc
    if synth_params is not None:
        arrays["synth"] = np.array(synth_params, dtype=np.int64)
    if store_sets:
        arrays["sets"] = np.stack(out["sets"])
    OUT.mkdir(parents=True, exist_ok=True)
    path = OUT / f"{name}.npz"
    np.savez_compressed(path, **arrays)
    print(f"{name}: n={rr.n} m={rr.m} nwords={rr.nwords} cases={len(out['kind'])} -> {path.stat().st_size} bytes")


# regenerated from a seed: cfg2, cfg3 and cfg5 of BASELINE.json (the last: seeds, topologies, lengths, per-node
# changes and set checksums of one start tree and two rounds of NNI / SPR / TBR at the HBM-resident size)
SYNTH = [(64, 10000, 6, 2, 12, 4), (500, 50000, 3, 1, 9, 8), (2000, 200000, 9, 1, 6, 0)]


def main():
    if ob.load_ref() is None:
        raise SystemExit("needs /root/reference (oracle/_ref/liblvbref.so)")
    only = set(sys.argv[1:])  # e.g. `gen_golden.py synth_2000x200000` regenerates just that file
    if only:
        for (n, m, seed, trees, moves, thr) in SYNTH:
            if f"synth_{n}x{m}" in only:
                rr = ob.RefRun(rows=synth.treelike_rows(n, m, seed), seed=seed)
                generate(f"synth_{n}x{m}", rr, n_trees=trees, n_moves=moves, store_sets=False, store_text=False,
                         synth_params=(n, m, seed), threads=thr)
                rr.close()
        return
    # 1. the reference's own test alignments, read by the reference's own reader
    for f in sorted(REFTESTS.glob("*.phy")):
        big = f.stat().st_size > 60000
        rr = ob.RefRun(path=str(f), fmt=0, seed=20240 + len(f.name))
        generate(f"ref_{f.stem}", rr, n_trees=2 if big else 4, n_moves=9 if big else 15,
                 store_sets=not big, store_text=True, threads=3 if "thread" in f.name or "stock" in f.name else 0)
        rr.close()
    # 2. edge shapes: m = 1, 15, 16, 17, 33 and every accepted symbol
    for m in (1, 15, 16, 17, 33):
        rows = synth.iupac_rows(7, m, 40 + m)
        rows[0] = bytes([ord("A")] * m)  # keep every column variable against an unambiguous row 0
        rows[1] = bytes([ord("C")] * m)
        rr = ob.RefRun(rows=rows, seed=7 + m)
        generate(f"edge_m{m}", rr, n_trees=3, n_moves=9, store_sets=True, store_text=True)
        rr.close()
    # 3. mid-size shapes: regenerated from a seed (checksum stored), lengths/changes/crc only
    for (n, m, seed, trees, moves, thr) in SYNTH:
        rows = synth.treelike_rows(n, m, seed)
        rr = ob.RefRun(rows=rows, seed=seed)
        generate(f"synth_{n}x{m}", rr, n_trees=trees, n_moves=moves, store_sets=False, store_text=False,
                 synth_params=(n, m, seed), threads=thr)
        rr.close()


if __name__ == "__main__":
    main()
