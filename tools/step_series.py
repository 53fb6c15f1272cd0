"""Per-step wall times of lvbgpu_propose_score from a cold start (diagnostic: clock ramp, one-off costs)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from lvb_amd import api, host
from tests.synth import treelike_rows

n, m, B = 500, 50000, 4096
rows, _ = host.prepare_alignment(treelike_rows(n, m, 3))
ctx = api.FitchContext(text_rows=rows)
tree = host.HostTree(n, seed=3001)
tree.upload(ctx)
ts = []
for i in range(300):
    t0 = time.perf_counter()
    ctx.propose_score(B, 1, 1000 + i)
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e6
print("first 12:", np.round(ts[:12], 1))
for a in range(0, 300, 25):
    print(a, "mean %.1f max %.1f" % (ts[a:a + 25].mean(), ts[a:a + 25].max()))
time.sleep(0.5)
ts2 = []
for i in range(50):
    t0 = time.perf_counter()
    ctx.propose_score(B, 1, 5000 + i)
    ts2.append(time.perf_counter() - t0)
print("after 0.5 s idle:", np.round(np.array(ts2[:10]) * 1e6, 1), "mean", np.mean(ts2) * 1e6)
