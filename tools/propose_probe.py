#!/usr/bin/env python3
"""Per-kind timing of the device neighbourhood generator (run under rocprofv3 --kernel-trace --stats)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
from lvb_amd import api, host
from tests import synth

n, m, B = 500, 2048, 4096   # few sites: the walk is short, the generator is what is timed
rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 3))
ctx = api.FitchContext(text_rows=rows)
tree = host.HostTree(n, seed=11)
tree.upload(ctx)
out = np.zeros(B, dtype=np.int64)
import time
for kind in (0, 1, 2):
    for rep in range(3):
        ctx._chk(ctx.lib.lvbgpu_propose_score(ctx.h, B, kind, 1234 + rep, out))
    t0 = time.perf_counter()
    for rep in range(20):
        ctx._chk(ctx.lib.lvbgpu_propose_score(ctx.h, B, kind, 99 + rep, out))
    print(f"kind {kind}: {(time.perf_counter() - t0) / 20 * 1e6:.0f} us per step (B={B}, {m} sites)")
