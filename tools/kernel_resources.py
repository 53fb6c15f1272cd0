"""Registers, scratch and occupancy of every kernel of the library as the compiler reports them (CPU: hipcc cross-compiles).
Run before and after a kernel change: the scoring walk must keep 8 waves per SIMD and no kernel may use scratch memory.

  python tools/kernel_resources.py
"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
KEYS = ("VGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]")
bad = 0
for src in ("fitch_kernels.hip", "propose_kernels.hip"):
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-x", "hip",
                          str(ROOT / "lvb_amd" / "csrc" / src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True).stderr
    cur, rows = None, {}
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
        for key in KEYS:
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m and cur:
                rows[cur].setdefault(key, m.group(1))
    for name, r in rows.items():
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        v = [r.get(k, "?") for k in KEYS]
        print(f"{dem[:100]:100s} VGPR {v[0]:>3s}  SGPR {v[1]:>3s}  scratch {v[2]:>3s}  waves/SIMD {v[3]}")
        bad += v[2] not in ("0", "?")
sys.exit(1 if bad else 0)
