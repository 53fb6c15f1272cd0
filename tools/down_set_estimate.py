"""How many row reads would a program need if it stopped at the LOWEST node above all of a move's rewrites?  (CPU)

A tree's Fitch length does not depend on where it is rooted.  With, for every node v of the resident tree, the set
down(v) of "everything that is not below v" (what v's parent edge sees from above) and the cached cost ec(v) of joining
up(v) with down(v), a candidate's length is
    L(T) + sum over recomputed nodes at or below m of (new - old changes) + changes(up'(m), down(m)) - ec(m)
where m is the lowest common ancestor (new topology) of the nodes the move rewires: nothing above m has to be walked.
A full program reads D + 3 rows (D dirty nodes); the truncated one D + 3 - depth(m) (+ 1 - 1: down(m) replaces the
root's own row and its clean child).

  python tools/down_set_estimate.py [B] [moves ...]
"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lvb_amd import host  # noqa: E402

n = 500
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
walks = [int(a) for a in sys.argv[2:]] or [75, 3000]
kinds = {"nni": 0, "spr": 1, "tbr": 2}


def depths(parent, root):
    d = np.full(len(parent), -1)
    d[root] = 0

    def dep(v):
        path = []
        while d[v] < 0:
            path.append(v)
            v = parent[v]
        base = d[v]
        for x in reversed(path):
            base += 1
            d[x] = base
        return base
    for v in range(len(parent)):
        dep(v)
    return d


def estimate(tree, kind, B):
    full = trunc = 0
    p0, l0, r0 = tree.arrays()
    for _ in range(B):
        e = tree.propose(kind)
        t2 = host.HostTree(left=l0, right=r0, root=tree.root)
        t2.apply(e)
        p, l, r = t2.arrays()
        d = depths(p, tree.root)
        nodes = [int(x["node"]) for x in e]
        # dirty set: rewired nodes and their ancestors below the root
        dirty = set()
        for v in nodes:
            while v != tree.root and v not in dirty:
                dirty.add(v)
                v = int(p[v])
        # m: the lowest node that is an ancestor-or-self of every rewired node
        def anc(v):
            out = []
            while v != tree.root:
                out.append(v)
                v = int(p[v])
            return out
        common = None
        for v in nodes:
            a = anc(v)
            common = a if common is None else [x for x in common if x in set(a)]
        dm = max((int(d[x]) for x in common), default=0)   # 0: the paths meet at the root only
        D = len(dirty)
        full += D + 3
        trunc += D + 3 - dm
        t2.close()
    return full / B, trunc / B


for w in walks:
    tree = host.HostTree(n, seed=5)
    for _ in range(w):
        tree.apply(tree.propose(1))
    for name, kind in kinds.items():
        f, t = estimate(tree, kind, B)
        print(f"walk {w:5d} {name}: full {f:6.2f} rows per candidate, truncated at the rewrites' common ancestor {t:6.2f}  "
              f"({100 * (1 - t / f):.1f} % fewer)", flush=True)
