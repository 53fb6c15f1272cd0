"""Run the REAL reference program (oracle/_ref/lvb_ref) on the input files of its own black-box
tests (test/src/COMMON/test_matrix_*, test_min_*: copied as data under ref_tests/blackbox/) and record
what it does: exit status, the FATAL ERROR line if any, else the numbers it prints for a pinned seed.
-> tests/golden/ref_blackbox.json

    python tests/golden/gen_blackbox.py
"""
import json
import re
import shutil
import subprocess
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
REFBIN = ROOT / "oracle" / "_ref" / "lvb_ref"
FILES = ROOT / "tests" / "golden" / "ref_tests" / "blackbox"
FMT = {"clustal": "clustal", "fasta": "fasta", "nexus": "nexus"}


def main():
    cases = []
    for f in sorted(FILES.glob("*.infile")):
        name = f.name[: -len(".infile")]
        fmt = next((v for k, v in FMT.items() if f"_{k}_" in name), "phylip")
        with tempfile.TemporaryDirectory() as d:
            shutil.copy(f, Path(d) / "infile")
            args = ["-f", fmt, "-s", "4242", "-p", "1"]
            p = subprocess.run([str(REFBIN), *args], cwd=d, capture_output=True, text=True, timeout=300)
            case = {"name": name, "infile": f.name, "format": fmt, "args": args, "exit_status": p.returncode}
            m = re.search(r"FATAL ERROR: *(.*)", p.stdout)
            if m:
                case["fatal"] = m.group(1).strip()
            else:
                case["expect"] = {k: int(re.search(pat, p.stdout).group(1)) for k, pat in
                                  (("rearrangements", r"Rearrangements evaluated: +(\d+)"),
                                   ("trees", r"Topologies recovered: +(\d+)"), ("score", r"Tree score: +(\d+)"))}
                case["expect"]["t0"] = re.search(r"SA Starting Temperature: +([0-9.]+)", p.stdout).group(1)
            cases.append(case)
            print(case)
    (ROOT / "tests" / "golden" / "ref_blackbox.json").write_text(json.dumps(
        {"_comment": "What the compiled reference program does on the inputs of its own black-box tests "
                     "(gen_blackbox.py). Output of the reference, not of this repository's code.", "cases": cases},
        indent=1) + "\n")


if __name__ == "__main__":
    main()
