"""Two candidates per wave: what the generating workgroups' pairs (GenArgs::pairs) are worth.  A device-built batch is
drawn with LVBGPU_PAIR on, its pairs read back (lvbgpu_debug_pairs), every candidate's program rebuilt on the host from the
reported move - the device emits the same tokens - and the tokens the pairs share counted; beside it what pairing the whole
batch in the full order of the programs read backwards would share (tools/shared_suffix_estimate.py).
    gpurun -- python tools/pair_quality.py [B] [moves before]"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["LVBGPU_PAIR"] = "64"
import numpy as np
from lvb_amd import api, host
from tests.synth import treelike_rows

n, m = 500, 50000
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
walk = int(sys.argv[2]) if len(sys.argv) > 2 else 75
rows, _ = host.prepare_alignment(treelike_rows(n, m, 3))
ctx = api.FitchContext(text_rows=rows)
tree = host.HostTree(n, seed=3001)
tree.upload(ctx)
for _ in range(walk):
    e = tree.propose(1); ctx.commit(e); tree.apply(e)


def suffix(a, b):
    k = 0
    while k < len(a) and k < len(b) and a[len(a) - 1 - k] == b[len(b) - 1 - k]:
        k += 1
    return k


for kind in (1, 0):
    lens = ctx.propose_score(B, kind, 11)
    pairs = ctx.last_pairs(0)
    progs = []
    for b in range(B):
        edits, info = ctx.proposal_edits(b)
        progs.append(tree.program(mode=0, edits=edits)["toks"])
    total = sum(len(p) for p in progs)
    seen = np.zeros(B, int)
    shared = alone = long_ = 0
    for a, b in pairs:
        seen[a] += 1
        if b == 0xFFFFFFFF:
            alone += 1
            continue
        seen[b] += 1
        assert a // 16 == b // 16, (a, b)           # a workgroup's sixteen
        if len(progs[a]) <= 64 and len(progs[b]) <= 64:
            shared += suffix(progs[a], progs[b])
        else:
            long_ += 1
    assert (seen == 1).all(), "every candidate walks exactly once"
    # what the workgroup's greedy matching should have found (longest shared end first among each sixteen)
    want = 0
    for w0 in range(0, B, 16):
        idx = list(range(w0, min(B, w0 + 16)))
        sh = {(i, j): (suffix(progs[i], progs[j]) if len(progs[i]) <= 64 and len(progs[j]) <= 64 else 0) for i in idx for j in idx if j > i}
        used = set()
        for (i, j), v in sorted(sh.items(), key=lambda kv: (-kv[1], -(16 * (kv[0][0] - w0) + kv[0][1] - w0))):
            if i not in used and j not in used:
                used |= {i, j}
                want += v
    full = sorted(range(B), key=lambda i: tuple(progs[i][::-1].tolist()))
    nb = [suffix(progs[full[i]], progs[full[i + 1]]) for i in range(B - 1)]
    used, best = np.zeros(B, bool), 0
    for i in sorted(range(B - 1), key=lambda i: -nb[i]):
        if not used[i] and not used[i + 1] and nb[i] > 0:
            used[i] = used[i + 1] = True
            best += nb[i]
    print(f"kind {kind} B={B} moves {walk}: mean tokens {total / B:.1f}; pairs {len(pairs)} ({alone} alone, {long_} with a program "
          f"of more than 64 tokens); row reads shared: device pairs {shared / total:.3f}, the same matching on the host "
          f"{want / total:.3f}, whole batch in full order {best / total:.3f}", flush=True)
