"""SURVEY.md 8f rank 2 on the device: the reference's own search trajectory with every tree length
computed by the HIP kernels (lvbhost_reference_search -> lvbgpu_score_batch / lvbgpu_commit).  Whole
runs must equal the real reference program's: the golden runs of tests/golden/ref_trajectories.json
(among them `-s 12345 -a 0` on the 100 x 1000 example => 337 893 rearrangements, score 4199, 54
topologies) and, where oracle/_ref travelled, a fresh run of oracle/_ref/lvb_ref on an alignment made
here."""
import hashlib
import json
import os
import re
import subprocess
from pathlib import Path

import pytest

from tests import synth

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
GOLD = ROOT / "tests" / "golden"
REFBIN = ROOT / "oracle" / "_ref" / "lvb_ref"
CASES = json.loads((GOLD / "ref_trajectories.json").read_text())["cases"]
# the reference's own test_matrix_*_length inputs in the other three formats (-f fasta|nexus|clustal)
BLACKBOX = [dict(infile="blackbox/" + c["infile"], seed=4242, algorithm=1, cooling="g", max_trees=0, format=c["format"],
                 expect=dict(c["expect"], outtree_sha256=None, outtree_head=None))
            for c in json.loads((GOLD / "ref_blackbox.json").read_text())["cases"] if "expect" in c]
PICK = {("stock_100x1000.phy", 12345, 0), ("stock_100x1000.phy", 12345, 1), ("stock_100x1000.phy", 2024, 2),
        ("test_treelength_4.phy", 4242, 0), ("test_treelength_4.phy", 4242, 1), ("test_treelength_4.phy", 4242, 2),
        ("test_treelength_4.phy", 5, 1), ("test_treelength_1.phy", 7, 1), ("test_treelength_6_thread_3.phy", 465380177, 1)}
if not os.environ.get("LVB_ALL_TRAJ"):
    CASES = [c for c in CASES if (c["infile"], c["seed"], c["algorithm"]) in PICK or c["max_trees"]]
# each of these is a 1.4 M-rearrangement run of a 5-taxon matrix (one latency-bound device step per accepted
# move, ~1 min): one format by default, all three with LVB_ALL_TRAJ=1
CASES = CASES + [b for b in BLACKBOX if os.environ.get("LVB_ALL_TRAJ") or b["format"] == "clustal"]


def search(path, seed, algorithm, cooling="g", max_batch=None, max_trees=0, fmt="phylip", device_moves_min=None):
    from lvb_amd import api, host
    names, rows = host.read_alignment(path, fmt)
    rows, min_len = host.prepare_alignment(rows)
    ctx = api.FitchContext(text_rows=rows)
    try:
        p = host.refsearch_defaults()
        p.seed, p.algorithm, p.cooling_schedule, p.min_len_tree = seed, algorithm, 0 if cooling == "g" else 1, min_len
        p.max_trees = max_trees
        if max_batch:
            p.max_batch = max_batch
        if device_moves_min is not None:
            p.device_moves_min = device_moves_min
        res, tree = host.reference_search(ctx.h, p)
        out = []
        for t in tree.best_trees():
            if t.root != 0:  # PrintTreestack re-roots at the first taxon (Treestack.c:402-403)
                t.apply(t.reroot_edits(0), 0)
            out.append(host.newick(t, names))
        return res, "".join(out).encode()
    finally:
        ctx.close()


@pytest.mark.parametrize("case", CASES, ids=[f"{Path(c['infile']).stem}-s{c['seed']}-a{c['algorithm']}{c['cooling']}" + (f"-t{c['max_trees']}" if c["max_trees"] else "") for c in CASES])
def test_golden_run_of_the_reference_program(case):
    res, trees = search(GOLD / "ref_tests" / case["infile"], case["seed"], case["algorithm"], case["cooling"],
                        max_trees=case["max_trees"], fmt=case.get("format", "phylip"))
    e = case["expect"]
    assert f"{res['t0']:.8f}" == e["t0"]
    assert (res["rearrangements"], res["best_length"], res["trees"]) == (e["rearrangements"], e["score"], e["trees"])
    if e["outtree_sha256"] is not None:
        assert hashlib.sha256(trees).hexdigest() == e["outtree_sha256"]
    print(f"\n{case['infile']} -s {case['seed']} -a {case['algorithm']}: {res['rearrangements']} rearrangements in "
          f"{res['seconds']:.2f} s, {res['device_steps']} device steps, {res['scored']} candidates scored")


@pytest.mark.parametrize("n,m,seed,alg", [(40, 3000, 321, 1), (64, 2000, 99, 0)])
def test_fresh_run_side_by_side_with_the_reference_binary(tmp_path, n, m, seed, alg):
    if not REFBIN.exists():
        pytest.skip("oracle/_ref/lvb_ref did not travel")
    rows = synth.treelike_rows(n, m, 1000 + n)
    infile = tmp_path / "infile"
    with open(infile, "w") as f:
        f.write(f"{n} {m}\n")
        for i, r in enumerate(rows):
            f.write(f"T{i:<9d}{r if isinstance(r, str) else r.decode()}\n")
    p = subprocess.run([str(REFBIN), "-s", str(seed), "-a", str(alg), "-p", "1"], cwd=tmp_path, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-1500:]
    want = {k: int(re.search(rf"{k}: +(\d+)", p.stdout).group(1))
            for k in ("Rearrangements evaluated", "Topologies recovered", "Tree score")}
    want_t0 = re.search(r"SA Starting Temperature: +([0-9.]+)", p.stdout).group(1)
    res, trees = search(infile, seed, alg)
    assert f"{res['t0']:.8f}" == want_t0
    assert (res["rearrangements"], res["trees"], res["best_length"]) == (
        want["Rearrangements evaluated"], want["Topologies recovered"], want["Tree score"])
    assert trees == (tmp_path / "outtree").read_bytes()


SYNTH = json.loads((GOLD / "ref_trajectories_synthetic.json").read_text())["cases"]


@pytest.mark.parametrize("case", SYNTH, ids=[f"synthetic-{c['taxa']}x{c['sites']}-s{c['seed']}-a{c['algorithm']}" for c in SYNTH])
def test_golden_run_at_a_shape_with_long_batches(tmp_path, case):
    """the small golden runs above are latency-bound chains of short batches; here a run at a
    batch-sized shape (200 x 20 000: 10 tiles, 31 000 device steps) whose batches go to the device as their 16-byte move
    parameters (lvbgpu_score_moves: refsearch.cpp's device-move path).  The reference's numbers and its
    output tree were recorded here by tests/golden/gen_ref_trajectories.py --synthetic (25-100 s of oracle/_ref/lvb_ref
    on the CPU); the alignment is regenerated from its seed."""
    from tests.golden.gen_ref_trajectories import write_synthetic
    infile = tmp_path / "infile"
    write_synthetic(infile, case["taxa"], case["sites"])
    # (this run's batches stay below the default threshold of 128 proposals for the device-move path - the chain accepts
    # too often - so the threshold is lowered: every batch of 8 proposals or more is scored from its move parameters)
    res, trees = search(infile, case["seed"], case["algorithm"], device_moves_min=8)
    e = case["expect"]
    assert f"{res['t0']:.8f}" == e["t0"]
    assert (res["rearrangements"], res["best_length"], res["trees"]) == (e["rearrangements"], e["score"], e["trees"])
    assert hashlib.sha256(trees).hexdigest() == e["outtree_sha256"]
    assert res["device_move_steps"] > 1000, "the device-move path (lvbgpu_score_moves) was hardly used: the shape no longer tests it"
    print(f"\n{case['taxa']} x {case['sites']} -s {case['seed']} -a {case['algorithm']}: {res['rearrangements']} rearrangements in "
          f"{res['seconds']:.2f} s, {res['device_steps']} device steps ({res['device_move_steps']} through lvbgpu_score_moves), "
          f"{res['scored']} candidates scored")
