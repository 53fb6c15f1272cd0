"""anneal_chains at 500 x 50k: R chains for a given time, the chains that did not freeze (and the first two) in detail.
Usage: chains_probe2.py R seconds"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lvb_amd import api, host
from tests.synth import treelike_rows
n, m = 500, 50000
rows, min_len = host.prepare_alignment(treelike_rows(n, m, 3))
R = int(sys.argv[1]); secs = float(sys.argv[2])
ctx = api.FitchContext(text_rows=rows)
trees = [host.HostTree(n, seed=300100 + c) for c in range(R)]
ps = []
for c in range(R):
    p = host.anneal_defaults()
    p.seed = 23757 + c + 1; p.algorithm = 11; p.batch = 4096; p.t0 = 0.0; p.min_len_tree = min_len
    p.max_seconds = secs; p.log_cap = 16
    ps.append(p)
res, log = host.anneal_chains(ctx, trees, ps)
for c, r in enumerate(res):
    if not r["frozen"] or c < 2:
        print(c, {k: r[k] for k in ("start_length", "best_length", "final_length", "consumed", "accepted", "temperatures", "device_steps", "scored", "frozen", "t_final")})
