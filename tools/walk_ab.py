"""A/B of the scoring walk at the bench shape: kernel-only replay of host-built (LPT) batches and device-built ones."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from lvb_amd import api, host
from tests.synth import treelike_rows

n, m, B = 500, 50000, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
walk = int(sys.argv[2]) if len(sys.argv) > 2 else 75
kind = int(sys.argv[3]) if len(sys.argv) > 3 else 1   # 0 NNI, 1 SPR, 2 TBR
rows, _ = host.prepare_alignment(treelike_rows(n, m, 3))
ctx = api.FitchContext(text_rows=rows)
tree = host.HostTree(n, seed=3001)
tree.upload(ctx)
for _ in range(walk):
    e = tree.propose(1); ctx.commit(e); tree.apply(e)
batches = []
for _ in range(4):
    offs, edits = tree.propose_batch(kind, B)
    bh = api.C.c_void_p()
    ctx._chk(ctx.lib.lvbgpu_batch_build(ctx.h, B, offs, edits.ctypes.data, None, api.C.byref(bh)))
    batches.append(api.Batch(ctx, bh, B))
ref = [None] * 4
for rep in range(3):
    for i, b in enumerate(batches):
        b.launch()
        l = b.lengths()
        assert ref[i] is None or np.array_equal(ref[i], l)
        ref[i] = l
for _ in range(20):
    for b in batches: b.launch()
ctx.synchronize()
ctx.timer_start()
K = 200
for i in range(K):
    batches[i % 4].launch()
ms = ctx.timer_stop()
st = batches[0].stats()
print(f"host-built LPT  B={B} D={st['dirty_nodes']/B:.1f}: {1e3*ms/K:.1f} us per launch, {st['algorithmic_bytes']/(ms/K*1e-3)/1e12:.2f} TB/s")
# device-built (unsorted)
ctx.walk_timing(1)
for i in range(100):
    ctx.propose_score(B, kind, 77 + i)
wms, k = ctx.walk_timing_read()
ctx.walk_timing(0)
print(f"device-built    B={B}: {1e3*wms/k:.1f} us per walk")
# ... and the whole pipelined step (generator [+ pair sort] + walk, two in flight), as bench.py's headline measures it
import time
draw = np.zeros(1, dtype=api.DRAW_DTYPE)
draw[0]["chain"], draw[0]["count"], draw[0]["kind"] = 0, B, kind
outs = [np.zeros(B, dtype=np.int64), np.zeros(B, dtype=np.int64)]
def submit(slot, seed):
    draw[0]["seed"] = seed
    ctx._chk(ctx.lib.lvbgpu_chains_submit(ctx.h, slot, 1, draw.ctypes.data))
def run(n, s0):
    for j in range(2):
        submit(j, s0 + j)
    for i in range(n):
        ctx._chk(ctx.lib.lvbgpu_chains_collect(ctx.h, i % 2, outs[i % 2]))
        if i + 2 < n:
            submit(i % 2, s0 + i + 2)
run(300, 10)
ctx.walk_timing(4)
t0 = time.perf_counter()
run(200, 1000)
dt = time.perf_counter() - t0
wms, k = ctx.walk_timing_read()
ctx.walk_timing(0)
print(f"pipelined step  B={B}: {1e6 * dt / 200:.1f} us per step = {B * 200 / dt / 1e6:.1f} M candidates/s, walk {1e3 * wms / k:.1f} us")
chk = ctx.propose_score(B, kind, 4242)
print("checksum", int(chk.sum()), int(chk.min()))
