"""anneal_chains at 500 x 50k for a few R: where does a step's time go (LVBHOST_PROFILE=1 prints the breakdown).
Arguments: numbers of chains, e.g.  1 32"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
if "--quiet" not in sys.argv:
    os.environ["LVBHOST_PROFILE"] = "1"
from lvb_amd import api, host
from tests.synth import treelike_rows

n, m = 500, 50000
rows, min_len = host.prepare_alignment(treelike_rows(n, m, 3))
for spec in [a for a in sys.argv[1:] if not a.startswith("--")] or ["1", "16"]:
    R = int(spec)
    ctx = api.FitchContext(text_rows=rows)
    trees = [host.HostTree(n, seed=(9 * 1000 + 1) * 100 + c) for c in range(R)]   # bench.py's default --anneal-seed 9: no straggler among 32
    ps = []
    for c in range(R):
        p = host.anneal_defaults()
        p.seed = 9 * 7919 + c + 1; p.algorithm = 11; p.batch = 4096; p.t0 = float(os.environ.get("PROBE_T0", "0")); p.min_len_tree = min_len
        p.max_seconds = 8.0; p.log_cap = 4096
        p.lanes = int(os.environ.get("PROBE_LANES", "0"))          # 0: the library's default (2 from 16 chains on)
        p.run_levels = int(os.environ.get("PROBE_RUN_LEVELS", "0"))   # one chain: runs of acceptances in one step (host-drawn hot phase)
        if os.environ.get("PROBE_NO_REROOT"):
            p.reroot_interval = 0   # upper bound of what cheaper re-roots could give (the trajectories change)
        ps.append(p)
    t0 = time.perf_counter()
    res, log = host.anneal_chains(ctx, trees, ps)
    dt = time.perf_counter() - t0
    # when the best length first got below a few marks (the reference program stands at ~3.9 M after 20 s)
    marks = {L: next((round(t, 3) for t, b in log if b <= L), None) for L in (5000000, 3900000, 3000000, 2600000)}
    print(f"R={R}: {dt:.3f} s, scored/s {sum(r['scored'] for r in res)/dt:.0f}, consumed/s {sum(r['consumed'] for r in res)/dt:.0f}, "
          f"steps {max(r['device_steps'] for r in res)}, best {min(r['best_length'] for r in res)}, frozen {sum(r['frozen'] for r in res)}, "
          f"seconds to length {marks}", flush=True)
    for t in trees:
        t.close()
    ctx.close()
