"""Same run, two programs: the reference binary (oracle/_ref/lvb_ref, CPU) and
`lvbhost_reference_search` (lengths from the MI355X) on one synthetic alignment and seed.  Prints one
JSON line: both wall times, the numbers both programs report, and whether the output trees are
byte-identical.  Needs oracle/_ref (built here by oracle/Makefile; it travels to the GPU box).

    python tests/manual/exact_vs_reference.py --taxa 200 --sites 20000 --seed 77 -a 1
"""
import argparse
import json
import re
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--taxa", type=int, default=200)
    ap.add_argument("--sites", type=int, default=20000)
    ap.add_argument("--seed", type=int, default=77)
    ap.add_argument("-a", dest="alg", type=int, default=1)
    ap.add_argument("--threads", type=int, default=1, help="-p of the reference run")
    ap.add_argument("--skip-reference", action="store_true")
    a = ap.parse_args()
    from lvb_amd import search
    from tests import synth
    rows = synth.treelike_rows(a.taxa, a.sites, 4000 + a.taxa)
    out = {"taxa": a.taxa, "sites": a.sites, "seed": a.seed, "algorithm": a.alg}
    with tempfile.TemporaryDirectory() as d:
        d = Path(d)
        with open(d / "infile", "w") as f:
            f.write(f"{a.taxa} {a.sites}\n")
            for i, r in enumerate(rows):
                f.write(f"T{i:<9d}{r if isinstance(r, str) else r.decode()}\n")
        t0 = time.perf_counter()
        res = search.run_exact(str(d / "infile"), a.seed, a.alg, out=str(d / "ours.tre"), verbose=False)
        out["gpu"] = {"wall_s": round(time.perf_counter() - t0, 3), "search_s": round(res["seconds"], 3),
                      "device_s": round(res["seconds_device"], 3), "rearrangements": res["rearrangements"],
                      "score": res["best_length"], "trees": res["trees"], "t0": f"{res['t0']:.8f}",
                      "device_steps": res["device_steps"], "scored": res["scored"],
                      "accepted_moves": res["accepted_moves"]}
        print(json.dumps(out), file=sys.stderr, flush=True)  # kept even if the reference run below is cut short
        if not a.skip_reference:
            t0 = time.perf_counter()
            p = subprocess.run([str(ROOT / "oracle" / "_ref" / "lvb_ref"), "-s", str(a.seed), "-a", str(a.alg), "-p",
                                str(a.threads)], cwd=d, capture_output=True, text=True)
            wall = time.perf_counter() - t0
            g = lambda pat: re.search(pat, p.stdout).group(1)
            out["reference"] = {"wall_s": round(wall, 3), "threads": a.threads,
                                "rearrangements": int(g(r"Rearrangements evaluated: +(\d+)")),
                                "score": int(g(r"Tree score: +(\d+)")), "trees": int(g(r"Topologies recovered: +(\d+)")),
                                "t0": g(r"SA Starting Temperature: +([0-9.]+)")}
            out["identical_output_trees"] = (d / "ours.tre").read_bytes() == (d / "outtree").read_bytes()
            out["same_numbers"] = all(out["gpu"][k] == out["reference"][k] for k in ("rearrangements", "score", "trees", "t0"))
            out["speedup"] = round(wall / out["gpu"]["wall_s"], 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
