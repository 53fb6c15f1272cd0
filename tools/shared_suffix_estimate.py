"""How many of a batch's row reads would walking two candidates per wave save?  (CPU, numpy)

Two candidates whose programs end alike - the same clean rows with the same token flags from some point to the root -
could be walked by ONE wave that carries two states (accumulator, operand stack, counters) and loads every row of the
common suffix once: per shared token one load instead of two (the combines stay two).

Estimate on the bench's own batches (500-taxon tree shape, B = 4096; start tree + 75 moves, and + 3000 moves), programs
from the host builder (the device generator emits the same programs).  Four orders of the candidates, consecutive ones
paired:
  full      sorted by the program read backwards (what a full sort gives; greedy matching of neighbours: "pairs")
  groups4   the same, four candidates per wave
  key       sorted by ONE small key a device could produce in a single counting-sort pass: the preorder number of the
            bottom node of the program's last chain
  plain     full order, but only suffixes without a chain start or merge inside (what a simple paired loop could take)
  window W  the candidates in the order of the DRAW, cut into runs of W (what one generator workgroup holds: W = 16; with
            several candidates per wave 32 or 64), each run sorted by the full order among its own and paired greedily:
            the pairing a workgroup could do by itself, without a pass over the whole batch

  python tools/shared_suffix_estimate.py [B] [moves]
"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lvb_amd import host  # noqa: E402

n = 500
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kinds = {"nni": 0, "spr": 1, "tbr": 2}
FRESH, PUSH, MSH, MMASK = 1 << 30, 1 << 31, 24, 0x3F


def plain(t):
    return (t & (FRESH | PUSH)) == 0 and ((t >> MSH) & MMASK) == 0


def common_suffix(a, b, only_plain=False):
    k, la, lb = 0, len(a), len(b)
    while k < la and k < lb and a[la - 1 - k] == b[lb - 1 - k] and (not only_plain or plain(int(a[la - 1 - k]))):
        k += 1
    return k


def preorder(tree):
    p, l, r = tree.arrays()
    tin, cnt, st = np.zeros(len(p), int), 0, [tree.root]
    while st:
        v = st.pop()
        tin[v] = cnt
        cnt += 1
        if l[v] >= 0:
            st.append(r[v])
            st.append(l[v])
    return tin


def last_chain_bottom(t, d, root):
    """the node the last chain of the program starts at (its second combine where the first is the re-used graft node)"""
    f = max(j for j in range(len(t)) if int(t[j]) & FRESH)
    comb = 0
    for j in range(f):
        tj = int(t[j])
        if not (tj & FRESH):
            comb += 1
        comb += (tj >> MSH) & MMASK
    cand = [x for x in d[comb:comb + 2] if x >= 0]
    return cand[1] if len(cand) > 1 else (cand[0] if cand else root)


def estimate(tree, kind, B):
    tin = preorder(tree)
    progs, keys = [], []
    for _ in range(B):
        pr = tree.program(mode=0, edits=tree.propose(kind))
        progs.append(pr["toks"])
        keys.append(tin[last_chain_bottom(pr["toks"], pr["dsts"], tree.root)])
    total = sum(len(p) for p in progs)
    full = sorted(range(B), key=lambda i: tuple(progs[i][::-1].tolist()))
    nb = [common_suffix(progs[full[i]], progs[full[i + 1]]) for i in range(B - 1)]
    used, pairs = np.zeros(B, bool), 0
    for i in sorted(range(B - 1), key=lambda i: -nb[i]):   # greedy matching on the path graph
        if not used[i] and not used[i + 1] and nb[i] > 0:
            used[i] = used[i + 1] = True
            pairs += nb[i]
    groups4 = sum(min(nb[g0:g0 + 3]) * 3 for g0 in range(0, B - 3, 4))
    by_key = sorted(range(B), key=lambda i: keys[i])
    consecutive = lambda order, only_plain=False: sum(common_suffix(progs[order[i]], progs[order[i + 1]], only_plain)
                                                      for i in range(0, B - 1, 2))
    windows = {}
    for W in (16, 32, 64, 256):
        got = 0
        for w0 in range(0, B, W):
            run = sorted(range(w0, min(B, w0 + W)), key=lambda i: tuple(progs[i][::-1].tolist()))
            nbw = [common_suffix(progs[run[i]], progs[run[i + 1]]) for i in range(len(run) - 1)]
            usedw = np.zeros(len(run), bool)
            for i in sorted(range(len(run) - 1), key=lambda i: -nbw[i]):
                if not usedw[i] and not usedw[i + 1] and nbw[i] > 0:
                    usedw[i] = usedw[i + 1] = True
                    got += nbw[i]
        windows[W] = got / total
    return (total / B, pairs / total, groups4 / total, consecutive(by_key) / total, consecutive(full, True) / total, windows)


for walk in ([int(sys.argv[2])] if len(sys.argv) > 2 else [75, 3075]):
    tree = host.HostTree(n, seed=3001)
    for _ in range(walk):
        tree.apply(tree.propose(1))
    for name, kind in kinds.items():
        mean_tok, pairs, groups4, key, pl, windows = estimate(tree, kind, B)
        print(f"walk {walk:5d} {name}: B={B} mean tokens {mean_tok:.1f} (D = {mean_tok - 3:.1f}); row reads shared: full order, pairs "
              f"{pairs:.3f}, groups of 4 {groups4:.3f}; one-pass key {key:.3f}; plain suffixes only {pl:.3f}; paired within runs of the draw: "
              + ", ".join(f"{W}: {v:.3f}" for W, v in windows.items()), flush=True)
    tree.close()
