"""How many of a batch's row reads would pairing candidates in one wave save?  (CPU, numpy)

Two candidates whose root-ward dirty paths run together from some junction node up to the root read the SAME clean rows
for that stretch (the siblings off the common path) with the same token flags.  A wave walking both programs would load
each of those rows once and feed two accumulators: per shared token one load instead of two (the combines stay two).

Estimate on the bench's own batches (500 x 50k tree shape, SPR, B = 4096; start tree + 75 moves, and + 3000 moves):
programs from the host builder (the device generator emits the same programs), candidates sorted by their program read
backwards (so neighbours in the order share the longest suffixes) and paired greedily: best of the two neighbours
first.  Reported: the fraction of all row reads that disappear, and the same restricted to suffixes of whole tokens
without merges (what a simple paired loop could take).

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


def common_suffix(a, b):
    k, la, lb = 0, len(a), len(b)
    while k < la and k < lb and a[la - 1 - k] == b[lb - 1 - k]:
        k += 1
    return k


def estimate(tree, kind, B, group=2):
    progs = []
    for _ in range(B):
        e = tree.propose(kind)
        progs.append(tree.program(mode=0, edits=e)["toks"])
    total = sum(len(p) for p in progs)
    order = sorted(range(B), key=lambda i: tuple(progs[i][::-1].tolist()))
    # suffix shared by neighbours in the sorted order
    nb = [common_suffix(progs[order[i]], progs[order[i + 1]]) for i in range(B - 1)]
    if group == 2:
        # greedy matching on the path graph: take the longest shared suffixes first
        used = np.zeros(B, bool)
        saved = 0
        for i in sorted(range(B - 1), key=lambda i: -nb[i]):
            if not used[i] and not used[i + 1] and nb[i] > 0:
                used[i] = used[i + 1] = True
                saved += nb[i]
        return total, saved, float(np.mean([len(p) for p in progs]))
    # groups of `group` consecutive candidates: everybody shares the group's common suffix with the first
    saved = 0
    for g0 in range(0, B - group + 1, group):
        k = min(nb[g0:g0 + group - 1])
        saved += k * (group - 1)
    return total, saved, float(np.mean([len(p) for p in progs]))


for walk in ([int(sys.argv[2])] if len(sys.argv) > 2 else [75, 3075]):
    tree = host.HostTree(n, seed=3001)
    for _ in range(walk):
        tree.apply(tree.propose(1))
    for name, kind in kinds.items():
        total, saved, mean_tok = estimate(tree, kind, B)
        t4, s4, _ = estimate(tree, kind, B, group=4)
        print(f"walk {walk:5d} {name}: B={B} mean tokens {mean_tok:.1f} (D = {mean_tok - 3:.1f}); pairs: {saved / total:.3f} of the row reads "
              f"shared; groups of 4: {s4 / t4:.3f}", flush=True)
    tree.close()
