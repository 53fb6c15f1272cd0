"""cfg5 (2000 x 200 000, TBR): scoring walk per launch, device-built batches of B candidates."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from lvb_amd import api, host
from tests.synth import treelike_rows
n, m = 2000, 200000
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rows, _ = host.prepare_alignment(treelike_rows(n, m, 3))
ctx = api.FitchContext(text_rows=rows)
del rows
tree = host.HostTree(n, seed=3001)
tree.upload(ctx)
for _ in range(30):
    e = tree.propose(2); ctx.commit(e); tree.apply(e)
for i in range(10):
    ctx.propose_score(B, 2, i)
ctx.walk_timing(1)
t0 = time.perf_counter()
K = 60
for i in range(K):
    ctx.propose_score(B, 2, 100 + i)
dt = time.perf_counter() - t0
wms, k = ctx.walk_timing_read()
st = ctx.proposal_stats()
print(f"B={B} D={st['dirty_nodes']/st['candidates']:.1f}: walk {1e3*wms/k:.1f} us, {st['algorithmic_bytes']/(wms/k*1e-3)/1e12:.2f} TB/s algorithmic, "
      f"step {1e6*dt/K:.1f} us, {B*K/dt/1e6:.2f} M/s")
