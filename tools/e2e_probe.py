#!/usr/bin/env python3
"""Where does an end-to-end scoring step spend its time? (GPU box)"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
from lvb_amd import api, host
from tests import synth

n, m = 500, 50000
rows, minlen = host.prepare_alignment(synth.treelike_rows(n, m, 3))
ctx = api.FitchContext(text_rows=rows)
tree = host.HostTree(n, seed=11)
tree.upload(ctx)
for B in (256, 1024, 4096, 16384):
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        offs, edits = tree.propose_batch(1, B)
    t_prop = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        b = ctx.build_batch.__func__  # noqa
        bh = api.C.c_void_p()
        ctx._chk(ctx.lib.lvbgpu_batch_build(ctx.h, B, offs, edits.ctypes.data, None, api.C.byref(bh)))
        bt = api.Batch(ctx, bh, B)
        bt.free()
    t_build = (time.perf_counter() - t0) / reps
    out = np.zeros(B, dtype=np.int64)
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx._chk(ctx.lib.lvbgpu_score_batch(ctx.h, B, offs, edits.ctypes.data, None, out))
    t_score = (time.perf_counter() - t0) / reps
    print(f"B={B}: propose {1e6*t_prop:.0f} us, batch_build(alloc+build+H2D) {1e6*t_build:.0f} us, "
          f"score_batch(build+H2D+kernel+D2H) {1e6*t_score:.0f} us -> {B/t_score/1e6:.2f} M/s e2e, "
          f"{B/(t_score+t_prop)/1e6:.2f} M/s incl. proposals")

print("neighbourhoods drawn on the device (lvbgpu_propose_score: draw + program + score + D2H of lengths/info):")
for B in (256, 1024, 4096, 16384, 65536):
    reps = 20
    ctx.propose_score(B, 1, 1)
    t0 = time.perf_counter()
    for r in range(reps):
        ctx.propose_score(B, 1, 100 + r)
    dt = (time.perf_counter() - t0) / reps
    print(f"B={B}: {1e6*dt:.0f} us/step -> {B/dt/1e6:.2f} M candidates/s end to end")

print("moves named by the host (lvbgpu_score_moves: 16 B per candidate up, the device builds rewrites + programs):")
rng = host.RefRng(5)
for B in (256, 1024, 4096, 16384):
    t0 = time.perf_counter()
    moves = np.array([tree.ref_draw_move(rng, 1) for _ in range(B)], dtype=api.MOVE_DTYPE)
    t_draw = time.perf_counter() - t0
    out = np.zeros(B, dtype=np.int64)
    for _ in range(3):
        ctx._chk(ctx.lib.lvbgpu_score_moves(ctx.h, B, moves.ctypes.data, out))
    t0 = time.perf_counter()
    for _ in range(20):
        ctx._chk(ctx.lib.lvbgpu_score_moves(ctx.h, B, moves.ctypes.data, out))
    dt = (time.perf_counter() - t0) / 20
    print(f"B={B}: {1e6*dt:.0f} us/step -> {B/dt/1e6:.2f} M candidates/s (drawing them in Python took {1e3*t_draw:.1f} ms)")
