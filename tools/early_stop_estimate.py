"""How many row reads would stopping a tile's walk save (where every recomputed set equals the old one)?"""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from lvb_amd import host
from tests.synth import treelike_rows

n, m = 500, 50000
walk = int(sys.argv[1]) if len(sys.argv) > 1 else 75
TILE = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
rows, _ = host.prepare_alignment(treelike_rows(n, m, 3))
code = {65: 1, 67: 2, 71: 4, 84: 8}
leaf = np.array([[code[c] for c in r] for r in rows], dtype=np.uint8)
m = leaf.shape[1]
ntiles = (m + TILE - 1) // TILE

def sets_of(left, right, root):
    S = np.zeros((2 * n - 3, m), np.uint8)
    S[:n] = leaf
    # post-order over internal nodes
    order, stack = [], [root]
    while stack:
        v = stack.pop()
        if left[v] >= 0:
            order.append(v); stack.append(left[v]); stack.append(right[v])
    for v in reversed(order):
        if v == root: continue
        a, b = S[left[v]], S[right[v]]
        x = a & b
        S[v] = np.where(x != 0, x, a | b)
    return S

tree = host.HostTree(n, seed=3001)
for _ in range(walk):
    e = tree.propose(1); tree.apply(e)
p, l, r = (a.copy() for a in tree.arrays()); root = tree.root
old = sets_of(l, r, root)
tot_full = tot_stop = tot_rec = tot_D = 0
for c in range(40):
    e = tree.propose(1)
    t2 = host.HostTree(left=l.copy(), right=r.copy(), root=root, seed=1)
    t2.apply(e)
    p2, l2, r2 = (a.copy() for a in t2.arrays())
    t2.close()
    new = sets_of(l2, r2, root)
    # dirty nodes = internal nodes (not root leaf) whose child pair changed or which have a dirty descendant
    changed_struct = {int(v) for v in range(n, 2 * n - 3) if (l[v], r[v]) != (l2[v], r2[v]) and {l[v], r[v]} != {l2[v], r2[v]}}
    dirty = set()
    for v in changed_struct:
        while v != root and v not in dirty:
            dirty.add(v); v = int(p2[v])
    D = len(dirty)
    # per tile: which dirty nodes must be recomputed with early stop
    diff = {v: np.array([(new[v, t * TILE:(t + 1) * TILE] != old[v, t * TILE:(t + 1) * TILE]).any() for t in range(ntiles)]) for v in dirty}
    # depth order (children before parents): sort by subtree size proxy = distance to root descending
    def depth(v):
        d = 0
        while v != root: v = int(p2[v]); d += 1
        return d
    need = {}
    rows_stop = np.zeros(ntiles)
    for v in sorted(dirty, key=depth, reverse=True):
        kids = [int(l2[v]), int(r2[v])]
        nd = np.zeros(ntiles, bool)
        if v in changed_struct:
            nd[:] = True
        for k in kids:
            if k in dirty:
                nd |= diff[k] & need[k] | (need[k] & diff[k])
        need[v] = nd
        clean = sum(1 for k in kids if k not in dirty)
        rows_stop += nd * clean
        # a recomputed node whose dirty child was NOT recomputed reads that child's old row
        for k in kids:
            if k in dirty:
                rows_stop += nd & ~need[k]
    # root combine: + root leaf + ... approximate as full: +3 - (shared) => use D+3 for full, rows_stop + 1 for stop
    tot_rec += sum(nd.sum() for nd in need.values()); tot_D += D * ntiles
    tot_full += (D + 3) * ntiles
    tot_stop += rows_stop.sum() + ntiles
print(f"recomputed nodes / dirty nodes = {tot_rec / tot_D:.3f}; with an old-row check at every recomputed node: {(tot_stop + tot_rec) / tot_full:.3f}, every 4th: {(tot_stop + tot_rec / 4 + 1.5 * tot_rec / max(tot_rec,1)) / tot_full:.3f}")
print(f"walk={walk} tile={TILE}: rows read with early stop / rows read now = {tot_stop / tot_full:.3f}")
