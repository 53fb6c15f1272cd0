#!/usr/bin/env python3
"""Time the batched SA host end to end on a synthetic alignment (GPU box)."""
import argparse, json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from lvb_amd import api, host
from tests import synth

ap = argparse.ArgumentParser()
ap.add_argument("--taxa", type=int, default=500); ap.add_argument("--sites", type=int, default=50000)
ap.add_argument("--batch", type=int, default=256); ap.add_argument("--seconds", type=float, default=5.0)
ap.add_argument("--alg", type=int, default=0); ap.add_argument("--t0", type=float, default=0.0)
ap.add_argument("--device-proposals", type=int, default=0)
a = ap.parse_args()
rows, minlen = host.prepare_alignment(synth.treelike_rows(a.taxa, a.sites, 3))
ctx = api.FitchContext(text_rows=rows)
tree = host.HostTree(a.taxa, seed=11)
print("start length", tree.upload(ctx))
p = host.anneal_defaults(); p.min_len_tree = minlen; p.batch = a.batch; p.algorithm = a.alg
p.device_proposals = a.device_proposals
t = time.perf_counter()
if a.t0 <= 0:
    t0 = host.starting_temperature(ctx, tree, p)
    print("t0", t0, "in", round(time.perf_counter() - t, 2), "s; length now", ctx.current_length())
else:
    t0 = a.t0
p.t0 = t0; p.max_seconds = a.seconds; p.log_cap = 1000
res, log = host.anneal(ctx, tree, p)
print(json.dumps(res))
print(log[:: max(1, len(log) // 15)])
