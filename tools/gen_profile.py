"""Where does a candidate's time in the generator go?  (LVBGPU_GEN_PROFILE: clock stamps of the first 256 candidates)"""
import ctypes as C, os, sys
from pathlib import Path
os.environ["LVBGPU_GEN_PROFILE"] = "1"
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from lvb_amd import api, host
from tests.synth import treelike_rows
n, m, B = 500, 50000, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows, _ = host.prepare_alignment(treelike_rows(n, m, 3))
ctx = api.FitchContext(text_rows=rows)
tree = host.HostTree(n, seed=3001)
tree.upload(ctx)
for _ in range(75):
    e = tree.propose(1); ctx.commit(e); tree.apply(e)
lib = ctx.lib
lib.lvbgpu_debug_generator_stamps.argtypes = [C.c_void_p, np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")]
for i in range(5):
    ctx.propose_score(B, kind, 100 + i)
st = np.zeros(256 * 8, dtype=np.uint64)
assert lib.lvbgpu_debug_generator_stamps(ctx.h, st) == 0
st = st.reshape(256, 8).astype(np.int64)
ok = st[:, 4] > 0
s = st[ok]
names = ["enter->tables in LDS", "tables->start", "draw", "rewrites", "program", "descriptors"]
cols = [s[:, 6] - s[:, 5], s[:, 0] - s[:, 6], s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 4] - s[:, 3]]
print(f"B={B} kind={kind}: clock ticks (s_memtime, 100 MHz = 10 ns each) per phase, mean / max over {ok.sum()} candidates")
for nme, c in zip(names, cols):
    print(f"  {nme:24s} {c.mean():8.1f} {c.max():8d}")
print(f"  {'total':24s} {(s[:, 4] - s[:, 5]).mean():8.1f} {(s[:, 4] - s[:, 5]).max():8d}")
if (s[:, 7] > 0).any():
    c = (s[:, 7] - s[:, 4])[s[:, 7] > 0]
    print(f"  {'descriptors->paired':24s} {c.mean():8.1f} {c.max():8d}   (waits for the workgroup's slowest candidate, then the pairing)")
    print(f"  {'enter->paired':24s} {(s[:, 7] - s[:, 5])[s[:, 7] > 0].mean():8.1f} {(s[:, 7] - s[:, 5])[s[:, 7] > 0].max():8d}")
print("  first candidate enters at", int(s[:, 5].min() - s[:, 5].min()), "last of the 256 ends at", int(s[:, 4].max() - s[:, 5].min()))
