"""Where do the multi-millisecond stalls of a stepping loop come from?  3000 steps, outliers listed."""
import gc, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from lvb_amd import api, host
from tests.synth import treelike_rows

n, m, B = 500, 50000, 4096
rows, _ = host.prepare_alignment(treelike_rows(n, m, 3))
ctx = api.FitchContext(text_rows=rows)
tree = host.HostTree(n, seed=3001)
tree.upload(ctx)
for mode in ("gc on", "gc off"):
    if mode == "gc off":
        gc.disable()
    out = np.zeros(B, dtype=np.int64)
    ts = np.zeros(3000)
    t_prev = time.perf_counter()
    for i in range(3000):
        ctx._chk(ctx.lib.lvbgpu_propose_score(ctx.h, B, 1, 1000 + i, out))
        t = time.perf_counter()
        ts[i] = t - t_prev
        t_prev = t
    us = ts * 1e6
    big = np.nonzero(us > 400)[0]
    print(mode, "median %.1f mean %.1f" % (np.median(us), us.mean()), "outliers:", [(int(i), round(float(us[i]))) for i in big][:20])
