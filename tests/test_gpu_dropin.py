"""Drop-in proof on the GPU: the reference PROGRAM (its own main, Anneal, mutate_*, treestack,
output code - compiled from /root/reference into oracle/_ref/lvb_dropin) linked against our
getplen adapter instead of its TreeEvaluation.o, so every tree length of the run is computed by the
HIP kernels.  It must reproduce the reference's own known answers and, run for run, the unmodified
reference binary (oracle/_ref/lvb_ref): same rearrangement count, score, topologies and output trees.
"""
import json
import re
import shutil
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
GOLD = ROOT / "tests" / "golden"
DROPIN = ROOT / "oracle" / "_ref" / "lvb_dropin"
REFBIN = ROOT / "oracle" / "_ref" / "lvb_ref"
CASES = json.loads((GOLD / "ref_tests.json").read_text())["cases"]
FIELDS = ("Rearrangements evaluated", "Topologies recovered", "Tree score")
# Every case passes (profiles/gpu_tests_r01_full_dropin.log: 19/19, 11 min - the 5- and 6-taxon runs
# make ~1.3 M strict-compat calls each).  By default the suite runs this subset (~3 min) and all
# of them with LVB_ALL_DROPIN=1.
import os  # noqa: E402

QUICK = {"test_treelength_1", "test_trees_recovered_1", "test_treelength_4", "test_trees_recovered_4",
         "test_treelength_5_thread_2", "test_treelength_6_thread_2", "test_treelength_6_thread_3",
         "test_treelength_7_thread_2", "stock_example_a0"}
if not os.environ.get("LVB_ALL_DROPIN"):
    CASES = [c for c in CASES if c["name"] in QUICK]


def _run(binary: Path, case: dict, workdir: Path) -> tuple[dict, bytes]:
    workdir.mkdir(parents=True, exist_ok=True)
    shutil.copy(GOLD / "ref_tests" / case["infile"], workdir / "infile")
    args = list(case["args"])
    if case["infile"].startswith("stock"):
        args = ["-i", "infile"] + args
    if "-s" not in args:
        args += ["-s", "4242"]  # the reference seeds from time() by default: pin it so two runs compare
    p = subprocess.run([str(binary), *args], cwd=workdir, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "FATAL ERROR" not in p.stdout
    out = {}
    for f in FIELDS + ("PThreads",):
        m = re.search(rf"{f}: +(\d+)", p.stdout)
        if m:
            out[f] = int(m.group(1))
    return out, (workdir / "outtree").read_bytes()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_reference_program_on_hip_getplen(case, tmp_path):
    if not DROPIN.exists() or not REFBIN.exists():
        pytest.skip("oracle/_ref binaries did not travel")
    got, trees = _run(DROPIN, case, tmp_path / "dropin")
    for key, want in case["expect"].items():
        assert got[key] == want, f"{case['name']}: {key} = {got.get(key)} (reference test expects {want})"
    # and the unmodified reference, run here with the same arguments, agrees line for line
    ref, ref_trees = _run(REFBIN, case, tmp_path / "ref")
    for f in FIELDS:
        assert got[f] == ref[f], f"{case['name']}: {f} differs from the reference binary"
    assert trees == ref_trees, "output trees differ from the reference binary's"


def test_adapter_fails_like_the_reference_without_a_device(tmp_path):
    """Error convention (Error.c:49-67): FATAL ERROR on stdout + exit status 1."""
    if not DROPIN.exists():
        pytest.skip("oracle/_ref/lvb_dropin did not travel")
    case = CASES[0]
    workdir = tmp_path / "nodev"
    workdir.mkdir()
    shutil.copy(GOLD / "ref_tests" / case["infile"], workdir / "infile")
    import os
    env = dict(os.environ, LVBGPU_DEVICE="99")
    p = subprocess.run([str(DROPIN)], cwd=workdir, capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode != 0
    assert "FATAL ERROR" in p.stdout
