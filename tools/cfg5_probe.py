"""What does the memory path deliver at cfg5's geometry (2000 x 200 000: a 401 MB tree block, 98 tiles, every tile's
column slice as large as an XCD's L2) with 4 and with 8 row loads in flight per wave?  Pure-load probe
(lvbgpu_probe_l2, LVBGPU_PROBE_VERBOSE prints both depths): if 8 in flight read no faster than 4 here, a deeper ring in
the walk would not either.   python tools/cfg5_probe.py [taxa sites]"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["LVBGPU_PROBE_VERBOSE"] = "1"
import numpy as np
from lvb_amd import api

n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2000, 200000)
nwords = api.words_per_row(m)
rng = np.random.default_rng(1)
enc = rng.integers(0, 2**63, size=(n, nwords), dtype=np.uint64) | np.uint64(0x1111111111111111)
ctx = api.FitchContext(enc)
for layout in ("tile-major (the resident layout)", "row-major (emulated: the layout until round 3)"):
    print("==", layout, file=sys.stderr, flush=True)
    if layout.startswith("row"):
        os.environ["LVBGPU_PROBE_ROW_MAJOR"] = "1"
    for B in ((1024, 4096) if n >= 1000 else (4096, 16384)):
        ctx.probe_l2(B, 34 if n >= 1000 else 24, 6)
ctx.close()
