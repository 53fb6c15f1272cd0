"""Test-side helpers: a numpy interpreter for the device's token programs and small topology
utilities.  Independent of both the product (lvb_amd) and the C oracle: the SWAR step is written
a third time here, per nibble, so a shared misreading cannot hide."""
from __future__ import annotations

import numpy as np

TOK_ROW_MASK = 0x00FFFFFF
TOK_MERGE_SHIFT = 24
TOK_MERGE_MASK = 0x3F
TOK_FRESH = 1 << 30
TOK_PUSH = 1 << 31


def _nibbles(words: np.ndarray) -> np.ndarray:
    """uint64[W] -> uint8[W*16] state sets, site order."""
    w = np.asarray(words, dtype=np.uint64)
    shifts = (np.arange(16, dtype=np.uint64) * np.uint64(4))
    return ((w[:, None] >> shifts[None, :]) & np.uint64(0xF)).astype(np.uint8).reshape(-1)


def _pack(nibs: np.ndarray) -> np.ndarray:
    n = nibs.reshape(-1, 16).astype(np.uint64)
    shifts = (np.arange(16, dtype=np.uint64) * np.uint64(4))
    return (n << shifts[None, :]).sum(axis=1, dtype=np.uint64)


def fitch_words(x: np.ndarray, y: np.ndarray):
    """Textbook Fitch per site on packed words -> (z words, number of unions)."""
    a, b = _nibbles(x), _nibbles(y)
    inter = a & b
    empty = inter == 0
    z = np.where(empty, a | b, inter)
    return _pack(z), int(empty.sum())


def run_program(toks, dsts, rows: np.ndarray):
    """Walk a token program the way the kernel does.  rows[r] = state-set words of row r.
    -> (total changes of all combines, {dst: (set words, changes)}, max stack depth used)."""
    acc = None
    stack = []
    out = {}
    total = 0
    k = 0
    depth = 0
    root_changes = 0

    def produced(z, ch):
        nonlocal k, total, root_changes
        d = int(dsts[k])
        k += 1
        total += ch
        if d >= 0:
            assert d not in out, "node produced twice"
            out[d] = (z, ch)
        else:
            root_changes += ch

    for tok in np.asarray(toks, dtype=np.uint64):
        tok = int(tok)
        row = rows[tok & TOK_ROW_MASK]
        if tok & TOK_FRESH:
            if tok & TOK_PUSH:
                assert acc is not None
                stack.append(acc)
                depth = max(depth, len(stack))
            acc = row
        else:
            assert not (tok & TOK_PUSH)
            acc, ch = fitch_words(acc, row)
            produced(acc, ch)
        for _ in range((tok >> TOK_MERGE_SHIFT) & TOK_MERGE_MASK):
            other = stack.pop()
            acc, ch = fitch_words(other, acc)
            produced(acc, ch)
    assert k == len(dsts), "combine count mismatch"
    return total, out, depth, root_changes


def apply_edits(left, right, edits):
    l, r = np.array(left, dtype=np.int64), np.array(right, dtype=np.int64)
    for e in edits:
        l[int(e["node"])] = int(e["left"])
        r[int(e["node"])] = int(e["right"])
    return l, r


def parents_of(left, right):
    p = np.full(len(left), -1, dtype=np.int64)
    for v in range(len(left)):
        if left[v] >= 0:
            p[left[v]] = v
            p[right[v]] = v
    return p


def edit_key(edits):
    """Order-insensitive identity of an edit list: {(node, frozenset(children))}."""
    return frozenset((int(e["node"]), frozenset((int(e["left"]), int(e["right"])))) for e in edits)


def splits_of(left, right, root, n):
    """Canonical bipartition set of an unrooted tree (taxon sets not containing taxon 0's side)."""
    left, right = np.asarray(left), np.asarray(right)
    memo = {}

    def leaves(v):
        if v in memo:
            return memo[v]
        if left[v] < 0:
            s = frozenset([v])
        else:
            s = leaves(int(left[v])) | leaves(int(right[v]))
        memo[v] = s
        return s

    import sys
    sys.setrecursionlimit(max(10000, 4 * len(left)))
    all_taxa = frozenset(range(n))
    out = set()
    stack = [int(left[root]), int(right[root])]
    while stack:
        v = stack.pop()
        if left[v] >= 0:
            s = leaves(v)
            comp = all_taxa - s
            side = s if 0 not in s else comp
            if 1 < len(side) < n - 1:
                out.add(side)
            stack.append(int(left[v]))
            stack.append(int(right[v]))
    return frozenset(out)
