"""anneal_chains at 500 x 50k for a few R: where does a step's time go (LVBHOST_PROFILE=1)."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["LVBHOST_PROFILE"] = "1"
from lvb_amd import api, host
from tests.synth import treelike_rows

n, m = 500, 50000
rows, min_len = host.prepare_alignment(treelike_rows(n, m, 3))
for R in [int(x) for x in (sys.argv[1:] or ["1", "16"])]:
    ctx = api.FitchContext(text_rows=rows)
    trees = [host.HostTree(n, seed=300100 + c) for c in range(R)]
    ps = []
    for c in range(R):
        p = host.anneal_defaults()
        p.seed = 23757 + c + 1; p.algorithm = 11; p.batch = 4096; p.t0 = 0.0; p.min_len_tree = min_len
        p.max_seconds = 8.0; p.log_cap = 16
        ps.append(p)
    t0 = time.perf_counter()
    res, log = host.anneal_chains(ctx, trees, ps)
    dt = time.perf_counter() - t0
    print(f"R={R}: {dt:.3f} s, scored/s {sum(r['scored'] for r in res)/dt:.0f}, consumed/s {sum(r['consumed'] for r in res)/dt:.0f}, "
          f"steps {max(r['device_steps'] for r in res)}, best {min(r['best_length'] for r in res)}, frozen {sum(r['frozen'] for r in res)}", flush=True)
    ctx.close()
