import os, sys
sys.path.insert(0, '.')
os.environ["LVBGPU_PROBE_VERBOSE"] = "1"
from lvb_amd import api, host
from tests.synth import treelike_rows
rows, _ = host.prepare_alignment(treelike_rows(500, 50000, 3))
ctx = api.FitchContext(text_rows=rows)
host.HostTree(500, seed=3001).upload(ctx)
for B in (4096, 16384):
    for rpw in (23, 46):
        print("B", B, "rows per wave", rpw, "->", round(ctx.probe_l2(B, rpw, 20)), "GB/s", flush=True)
