"""Batched annealing and the reference's trajectory at a given shape under the current thresholds (GPU box).
    python tools/search_probe.py [taxa sites]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lvb_amd import api, host  # noqa: E402
from tests import synth  # noqa: E402

n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (500, 50000)
rows, minlen = host.prepare_alignment(synth.treelike_rows(n, m, 3))
ctx = api.FitchContext(text_rows=rows)
for seed in (5, 6):
    tree = host.HostTree(n, seed=seed)
    tree.upload(ctx)
    p = host.anneal_defaults()
    p.seed = seed
    p.min_len_tree = minlen
    t0 = time.perf_counter()
    res, _ = host.anneal(ctx, tree, p)
    print(f"anneal seed {seed}: {time.perf_counter() - t0:.3f} s wall, best {res['best_length']}, "
          f"{res['device_steps']} steps, {res['scored']} scored", flush=True)
    tree.close()
for seed in (77,):
    p = host.refsearch_defaults()
    p.seed = seed
    p.algorithm = 1
    p.min_len_tree = minlen
    t0 = time.perf_counter()
    res, tree = host.reference_search(ctx.h, p)
    print(f"exact seed {seed}: {time.perf_counter() - t0:.3f} s wall, {res['rearrangements']} rearrangements, "
          f"score {res['best_length']}, {res['device_steps']} steps ({res['device_move_steps']} from moves)", flush=True)
    tree.close()
ctx.close()
