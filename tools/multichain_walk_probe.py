"""The same number of candidates in one step, drawn from ONE chain or spread over R chains: what does the scoring walk
cost?  (500 x 50 000, SPR; R x (total / R) candidates; walk duration from HIP events, lvbgpu_walk_timing)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from lvb_amd import api, host
from tests.synth import treelike_rows

n, m = 500, 50000
total = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
rows, _ = host.prepare_alignment(treelike_rows(n, m, 3))
for R in (1, 4, 16, 32, 64):
    ctx = api.FitchContext(text_rows=rows)
    ctx.set_chains(R)
    for c in range(R):
        t = host.HostTree(n, seed=500 + c)
        ctx.select_chain(c)
        t.upload(ctx)
        t.close()
    per = total // R
    draws = [(c, per, 1, 100 + c) for c in range(R)]
    for _ in range(30):
        ctx.chains_propose_score(draws)
    ctx.walk_timing(1)
    t0 = time.perf_counter()
    steps = 100
    for s in range(steps):
        ctx.chains_propose_score([(c, per, 1, 1000 * s + c) for c in range(R)])
    dt = time.perf_counter() - t0
    ms, k = ctx.walk_timing_read()
    ctx.walk_timing(0)
    print(f"R={R:2d} x {per:5d} candidates: step {1e6 * dt / steps:7.1f} us, walk {1e3 * ms / k:7.1f} us ({per * R / (ms / k) / 1e3:.1f} M candidates/s in the walk)", flush=True)
    ctx.close()
