"""Generate tests/golden/ref_trajectories.json: whole runs of the REAL reference program
(oracle/_ref/lvb_ref, compiled from /root/reference by oracle/Makefile) with pinned seeds, for the
reference-trajectory search (lvb_amd/csrc/refsearch.cpp) to reproduce.  Data only: the command line,
the four numbers the program prints and its output trees (sha256 + the first lines).

    python tests/golden/gen_ref_trajectories.py
"""
import hashlib
import json
import re
import shutil
import subprocess
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
REFBIN = ROOT / "oracle" / "_ref" / "lvb_ref"
FILES = ROOT / "tests" / "golden" / "ref_tests"

# (alignment, seed, -a, -c)
RUNS = [(f, s, a, "g") for f in ("test_treelength_1.phy", "test_treelength_2.phy", "test_treelength_3.phy",
                                   "test_treelength_4.phy") for s in (4242, 7) for a in (0, 1, 2)]
RUNS += [("test_treelength_5_thread_2.phy", 509739986, a, "g") for a in (0, 1, 2)]
RUNS += [("test_treelength_6_thread_2.phy", 509739986, a, "g") for a in (0, 1)]
RUNS += [("test_treelength_6_thread_3.phy", 465380177, 1, "g"), ("test_treelength_7_thread_2.phy", 99, 1, "g")]
RUNS += [("test_matrix_phylip_length.phy", 31337, 1, "g")]
RUNS += [("stock_100x1000.phy", 12345, 0, "g"), ("stock_100x1000.phy", 12345, 1, "g"),
         ("stock_100x1000.phy", 2024, 2, "g"),
         # linear cooling only where t0 is tiny: its gradient is 1e-10 per step (Solve.c:178, 430), so
         # from t0 = 0.15 the reference needs ~1.5e9 temperatures to freeze
         ("test_treelength_4.phy", 5, 1, "l")]
# -t N: stop as soon as the treestack holds N trees (Solve.c:447-450)
MAX_TREES = {("test_treelength_4.phy", 4242, 1, "g"): 10, ("stock_100x1000.phy", 12345, 0, "g"): 3}
RUNS = [r + (0,) for r in RUNS] + [k + (v,) for k, v in MAX_TREES.items()]

FIELDS = {"rearrangements": r"Rearrangements evaluated: +(\d+)", "trees": r"Topologies recovered: +(\d+)",
          "score": r"Tree score: +(\d+)", "t0": r"SA Starting Temperature: +([0-9.]+)"}


# A run at a shape where the speculative batches are LONG (hundreds of proposals drawn ahead, scored through
# lvbgpu_score_moves: refsearch.cpp's device-move path), on a synthetic alignment the test regenerates from its seed
# (tests/synth.treelike_rows(taxa, sites, 4000 + taxa), written as the reference's PHYLIP reader wants it)
SYNTHETIC = [(200, 20000, 77, 1)]


def write_synthetic(path, taxa, sites):
    import sys
    sys.path.insert(0, str(ROOT))
    from tests import synth
    rows = synth.treelike_rows(taxa, sites, 4000 + taxa)
    with open(path, "w") as f:
        f.write(f"{taxa} {sites}\n")
        for i, r in enumerate(rows):
            f.write(f"T{i:<9d}{r.decode()}\n")


def synthetic():
    cases = []
    for taxa, sites, seed, alg in SYNTHETIC:
        with tempfile.TemporaryDirectory() as d:
            write_synthetic(Path(d) / "infile", taxa, sites)
            args = ["-s", str(seed), "-a", str(alg), "-p", "1"]
            p = subprocess.run([str(REFBIN), *args], cwd=d, capture_output=True, text=True, timeout=3000, check=True)
            out = {}
            for k, pat in FIELDS.items():
                m = re.search(pat, p.stdout)
                out[k] = m.group(1) if k == "t0" else int(m.group(1))
            trees = (Path(d) / "outtree").read_bytes()
            out["outtree_sha256"] = hashlib.sha256(trees).hexdigest()
            cases.append({"taxa": taxa, "sites": sites, "seed": seed, "algorithm": alg, "args": args, "expect": out})
            print(taxa, sites, seed, alg, out)
    doc = {"_comment": "Runs of the compiled reference program (oracle/_ref/lvb_ref) on synthetic alignments "
                       "(tests/synth.treelike_rows(taxa, sites, 4000 + taxa)); see gen_ref_trajectories.py --synthetic. "
                       "Output of the reference, not of this repository's code.",
           "cases": cases}
    (ROOT / "tests" / "golden" / "ref_trajectories_synthetic.json").write_text(json.dumps(doc, indent=1) + "\n")


def main():
    import sys
    if "--synthetic" in sys.argv:
        return synthetic()
    cases = []
    for infile, seed, alg, cool, max_trees in RUNS:
        with tempfile.TemporaryDirectory() as d:
            shutil.copy(FILES / infile, Path(d) / "infile")
            args = ["-s", str(seed), "-a", str(alg), "-c", cool, "-p", "1"] + (["-t", str(max_trees)] if max_trees else [])
            p = subprocess.run([str(REFBIN), *args], cwd=d, capture_output=True, text=True, timeout=300, check=True)
            out = {}
            for k, pat in FIELDS.items():
                m = re.search(pat, p.stdout)
                out[k] = m.group(1) if k == "t0" else int(m.group(1))
            trees = (Path(d) / "outtree").read_bytes()
            out["outtree_sha256"] = hashlib.sha256(trees).hexdigest()
            out["outtree_head"] = trees.decode().splitlines()[:2]
            cases.append({"infile": infile, "seed": seed, "algorithm": alg, "cooling": cool, "max_trees": max_trees,
                          "args": args, "expect": out})
            print(cases[-1]["infile"], seed, alg, cool, max_trees, out["rearrangements"], out["score"], out["trees"], out["t0"])
    doc = {"_comment": "Runs of the compiled reference program (oracle/_ref/lvb_ref) with pinned seeds; see "
                       "gen_ref_trajectories.py. Output of the reference, not of this repository's code.",
           "cases": cases}
    (ROOT / "tests" / "golden" / "ref_trajectories.json").write_text(json.dumps(doc, indent=1) + "\n")


if __name__ == "__main__":
    main()
