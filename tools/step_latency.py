"""Latency of one small lvbgpu_score_batch step (the unit an exact-trajectory search pays per accepted move),
with and without direct steps (LVBGPU_DIRECT_STEPS).  Run on the GPU box: python tools/step_latency.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lvb_amd import api, host  # noqa: E402
from tests import synth  # noqa: E402


def run(n, m, B, direct, reps=3000):
    os.environ["LVBGPU_DIRECT_STEPS"] = "1" if direct else "0"
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 5))
    ctx = api.FitchContext(text_rows=rows)
    tree = host.HostTree(n, seed=9)
    _, left, right = tree.arrays()
    ctx.set_tree(left, right, 0)
    cands = [tree.propose(1 + (i % 2)) for i in range(B)]
    first = ctx.score_batch(cands)
    t0 = time.perf_counter()
    for _ in range(reps):
        got = ctx.score_batch(cands)
    dt = (time.perf_counter() - t0) / reps
    assert np.array_equal(got, first)
    ctx.close()
    return dt * 1e6, first


if __name__ == "__main__":
    for n, m in ((100, 1000), (200, 20000), (500, 50000)):
        for B in (1, 8, 64, 200):
            a, la = run(n, m, B, False)
            b, lb = run(n, m, B, True)
            assert np.array_equal(la, lb), "direct steps changed the lengths"
            print(f"{n}x{m} B={B:4d}: copy path {a:7.1f} us/step   direct {b:7.1f} us/step", flush=True)
