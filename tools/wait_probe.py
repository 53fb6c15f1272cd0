"""Where does a call queued behind a long kernel spend its time?  (lvbgpu_debug_stall + a wait limit)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from lvb_amd import api, host
from tests.synth import treelike_rows

rows, _ = host.prepare_alignment(treelike_rows(30, 900, 5))
ctx = api.FitchContext(text_rows=rows)
tree = host.HostTree(30, seed=6)
tree.upload(ctx)
ctx.propose_score(64, -1, 11)
ctx.set_wait_limit(0.05)
lib, h = ctx.lib, ctx.h
draw = np.zeros(1, dtype=api.DRAW_DTYPE)
draw[0]["chain"], draw[0]["count"], draw[0]["kind"], draw[0]["seed"] = 0, 64, -1, 11
out = np.zeros(64, dtype=np.int64)
for rnd in range(2):
    t0 = time.perf_counter()
    rc0 = lib.lvbgpu_debug_stall(h, 400)
    t1 = time.perf_counter()
    rc1 = lib.lvbgpu_chains_submit(h, 0, 1, draw.ctypes.data)
    t2 = time.perf_counter()
    rc2 = lib.lvbgpu_chains_collect(h, 0, out)
    t3 = time.perf_counter()
    print(f"stall launch {1e3*(t1-t0):.2f} ms rc {rc0} | submit {1e3*(t2-t1):.2f} ms rc {rc1} | collect {1e3*(t3-t2):.2f} ms rc {rc2} {ctx.last_error()[:80]}")
    ts = time.perf_counter()
    ctx.synchronize()
    print(f"  synchronize {1e3*(time.perf_counter()-ts):.2f} ms")
