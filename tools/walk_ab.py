"""A/B of the scoring walk at the bench shape: kernel-only replay of host-built (LPT) batches and device-built ones."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from lvb_amd import api, host
from tests.synth import treelike_rows

n, m, B = 500, 50000, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
walk = int(sys.argv[2]) if len(sys.argv) > 2 else 75
rows, _ = host.prepare_alignment(treelike_rows(n, m, 3))
ctx = api.FitchContext(text_rows=rows)
tree = host.HostTree(n, seed=3001)
tree.upload(ctx)
for _ in range(walk):
    e = tree.propose(1); ctx.commit(e); tree.apply(e)
batches = []
for _ in range(4):
    offs, edits = tree.propose_batch(1, B)
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
    ctx.propose_score(B, 1, 77 + i)
wms, k = ctx.walk_timing_read()
ctx.walk_timing(0)
print(f"device-built    B={B}: {1e3*wms/k:.1f} us per walk")
