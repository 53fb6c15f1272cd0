"""The reference-trajectory search (lvb_amd/csrc/refsearch.cpp) as HOST LOGIC, in the CPU tier: the
host sources are linked against a test double of the scorer (tests/cpu_double: the oracle behind the
few lvbgpu_* calls the host makes) and must reproduce whole runs of the real reference program -
rearrangement count, score, treestack size, starting temperature and the output trees byte for byte
(tests/golden/ref_trajectories.json, made by gen_ref_trajectories.py from oracle/_ref/lvb_ref).
The same runs go through the HIP scorer in tests/test_gpu_reftraj.py."""
import hashlib
import json
from pathlib import Path

import pytest

GOLD = Path(__file__).resolve().parent / "golden"
CASES = json.loads((GOLD / "ref_trajectories.json").read_text())["cases"]
# the reference's own test_matrix_*_length inputs in the other three formats (-f fasta|nexus|clustal)
BLACKBOX = [dict(infile="blackbox/" + c["infile"], seed=4242, algorithm=1, cooling="g", max_trees=0, format=c["format"],
                 expect=dict(c["expect"], outtree_sha256=None, outtree_head=None))
            for c in json.loads((GOLD / "ref_blackbox.json").read_text())["cases"] if "expect" in c]
import os  # noqa: E402

# every case passes (LVB_ALL_TRAJ=1, ~3 min on the CPU double); by default the tier runs the runs
# shorter than 400 k rearrangements, which include the 100-taxon example under all three -a schedules
QUICK = CASES + BLACKBOX if os.environ.get("LVB_ALL_TRAJ") else [c for c in CASES if c["expect"]["rearrangements"] < 400000]


def run_case(case, lib, new_ctx, free_ctx, max_batch=None):
    from lvb_amd import host
    names, rows = host.read_alignment(GOLD / "ref_tests" / case["infile"], case.get("format", "phylip"))
    rows, min_len = host.prepare_alignment(rows)
    ctx = new_ctx(rows)
    try:
        p = host.refsearch_defaults(lib)
        p.seed, p.algorithm = case["seed"], case["algorithm"]
        p.cooling_schedule = 0 if case["cooling"] == "g" else 1
        p.min_len_tree = min_len
        p.max_trees = case.get("max_trees", 0)
        p.device_moves_min = -1  # the test double has no device to build programs on
        if max_batch:
            p.max_batch = max_batch
        res, tree = host.reference_search(ctx, p, lib)
        out = []
        for t in tree.best_trees():
            if t.root != 0:  # PrintTreestack re-roots at the first taxon (Treestack.c:402-403)
                t.apply(t.reroot_edits(0), 0)
            out.append(host.newick(t, names))
        return res, "".join(out).encode()
    finally:
        free_ctx(ctx)


@pytest.fixture(scope="module")
def double():
    from oracle import binding
    binding.load_oracle()
    from tests.cpu_double import build
    return build.load()


def check(case, res, trees):
    e = case["expect"]
    assert f"{res['t0']:.8f}" == e["t0"]
    assert (res["rearrangements"], res["best_length"], res["trees"]) == (e["rearrangements"], e["score"], e["trees"])
    if e["outtree_sha256"] is not None:
        assert trees.decode().splitlines()[:2] == e["outtree_head"]
        assert hashlib.sha256(trees).hexdigest() == e["outtree_sha256"]


@pytest.mark.parametrize("case", QUICK, ids=[f"{Path(c['infile']).stem}-s{c['seed']}-a{c['algorithm']}{c['cooling']}" + (f"-t{c['max_trees']}" if c["max_trees"] else "") for c in QUICK])
def test_whole_run_matches_the_reference_program(double, case):
    lib, new_ctx, free_ctx = double
    res, trees = run_case(case, lib, new_ctx, free_ctx)
    check(case, res, trees)


@pytest.mark.parametrize("max_batch", [1, 3, 64, 999])
def test_trajectory_does_not_depend_on_how_far_ahead_proposals_are_drawn(double, max_batch):
    lib, new_ctx, free_ctx = double
    case = next(c for c in CASES if c["infile"] == "test_treelength_4.phy" and c["algorithm"] == 1 and c["cooling"] == "g"
                and not c["max_trees"])
    res, trees = run_case(case, lib, new_ctx, free_ctx, max_batch=max_batch)
    check(case, res, trees)
