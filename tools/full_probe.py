#!/usr/bin/env python3
"""Throughput of whole-tree evaluation (every internal node recomputed) - BASELINE.md 'full' mode."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
from lvb_amd import api, host
from tests import synth

for (n, m, B) in ((64, 10000, 1024), (500, 50000, 256), (500, 50000, 1024)):
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 3))
    ctx = api.FitchContext(text_rows=rows)
    lefts, rights = [], []
    for b in range(B):
        t = host.HostTree(n, seed=100 + b)
        _, l, r = t.arrays()
        lefts.append(l); rights.append(r); t.close()
    L, R = np.stack(lefts), np.stack(rights)
    out = ctx.score_full_batch(L, R)          # warm (includes host program build)
    t0 = time.perf_counter(); out2 = ctx.score_full_batch(L, R); dt = time.perf_counter() - t0
    assert (out == out2).all()
    # kernel-only: time a second launch through events is not exposed for full batches; report end-to-end
    bytes_alg = B * n * ctx.nwords * 8
    print(f"{n}x{m} B={B}: {dt*1e3:.2f} ms end to end (host program build + H2D + kernel + D2H) -> "
          f"{B/dt/1e3:.1f} k full evals/s, algorithmic {bytes_alg/dt/1e9:.0f} GB/s")
    ctx.close()
