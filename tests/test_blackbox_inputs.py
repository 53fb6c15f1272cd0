"""The input files of the reference's own black-box tests (test/src/COMMON/test_matrix_*, test_min_*;
data copied under tests/golden/ref_tests/blackbox/) through our reader and alignment preparation:
where the reference program stops with FATAL ERROR we must stop too (PHYLIP: with the same message), where it runs
we must hand the scorer the same matrix.  What the reference does is recorded in
tests/golden/ref_blackbox.json (gen_blackbox.py runs the compiled reference)."""
import json
from pathlib import Path

import pytest

GOLD = Path(__file__).resolve().parent / "golden"
CASES = json.loads((GOLD / "ref_blackbox.json").read_text())["cases"]


def load(case):
    """Main.c:88-100 as far as the scorer's input: read, size checks (Wrapper.c:53-63), matchange."""
    from lvb_amd import host
    names, rows = host.read_alignment(GOLD / "ref_tests" / "blackbox" / case["infile"], case["format"])
    if len(rows) < 5:
        raise ValueError("The data matrix must have at least 5 sequences.")
    return names, host.prepare_alignment(rows)


@pytest.mark.parametrize("case", [c for c in CASES if "fatal" in c], ids=lambda c: c["name"])
def test_inputs_the_reference_rejects_are_rejected_with_its_message(case):
    with pytest.raises(ValueError) as ei:
        load(case)
    ours = str(ei.value).splitlines()[0]
    # the reference names the file as it was given on its command line ("infile")
    ours = ours.replace(str(GOLD / "ref_tests" / "blackbox" / case["infile"]), "infile")
    if case["format"] == "phylip":
        assert ours == case["fatal"]
    else:
        # FASTA / NEXUS / CLUSTAL go through our own record-stream parser: the file must be refused where the
        # reference refuses it (its black-box tests look for "FATAL ERROR" only); the wording is ours
        assert ours


@pytest.mark.parametrize("case", [c for c in CASES if "expect" in c], ids=lambda c: c["name"])
def test_inputs_the_reference_accepts_give_its_matrix(case):
    from oracle import binding
    if binding.load_ref() is None:
        pytest.skip("oracle/_ref/liblvbref.so not available")
    names, (rows, min_len) = load(case)
    code = {"phylip": 0, "fasta": 1, "nexus": 2, "clustal": 3}[case["format"]]
    rr = binding.RefRun(path=str(GOLD / "ref_tests" / "blackbox" / case["infile"]), fmt=code, seed=1)
    try:
        assert rows == rr.rows() and min_len == rr.min_len
        assert [n.rstrip() for n in names] == [t.rstrip() for t in rr.titles()]
    finally:
        rr.close()
