"""Device time of an accepted move (lvbgpu_commit) and of a small scoring step at the bench shape.
Run on the GPU box: python tools/commit_latency.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lvb_amd import api, host  # noqa: E402
from tests import synth  # noqa: E402


def main():
    for n, m in ((100, 1000), (200, 20000), (500, 50000)):
        rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 5))
        ctx = api.FitchContext(text_rows=rows)
        tree = host.HostTree(n, seed=9)
        _, left, right = tree.arrays()
        ctx.set_tree(left, right, 0)
        reps = 300
        t_commit = 0.0
        for i in range(reps):
            e = tree.propose(1 + (i % 2))
            want = ctx.score_batch([e])[0]
            ctx.synchronize()
            t0 = time.perf_counter()
            got = ctx.commit(e)          # synchronous form: commit walk + read-back of the length
            t_commit += time.perf_counter() - t0
            tree.apply(e)
            assert got == want
        print(f"{n}x{m}: commit {t_commit / reps * 1e6:7.1f} us (lvbgpu_commit with its length read back; LVBGPU_DEFER_SLOTS=1 ~ one store per combine)", flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
