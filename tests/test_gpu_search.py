"""End to end through OUR host (reader -> cut -> device encode -> batched SA -> treestack mirror ->
tree writer): the reference's black-box known answers must come out (tests/golden/ref_tests.json).
The scorer is bit-exact (other tests); this checks the search mirror drives it to the same optima."""
import json
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

GOLD = Path(__file__).resolve().parent / "golden"
WANT = {c["infile"]: c["expect"] for c in json.loads((GOLD / "ref_tests.json").read_text())["cases"]}


@pytest.mark.parametrize("phy,score,topologies,on_device", [
    ("test_treelength_5_thread_2.phy", 1846, None, 0),
    ("test_treelength_6_thread_2.phy", 1628, 1, 0),
    ("test_treelength_6_thread_2.phy", 1628, 1, 1),
    ("test_treelength_7_thread_2.phy", 1006, 1, 1),
    ("test_treelength_6_thread_3.phy", 297, None, 2),
    ("test_treelength_7_thread_2.phy", 1006, 1, 0),
    ("test_treelength_5_thread_2.phy", 1846, None, 1),
], ids=["5t2-host", "6t2-host", "6t2-device", "7t2-device", "6t3-auto", "7t2-host-a2", "5t2-device-a2"])
def test_search_reaches_reference_optimum(tmp_path, phy, score, topologies, on_device, request):
    from lvb_amd import search
    out = tmp_path / "outtree"
    algorithm = 2 if request.node.callspec.id.endswith("a2") else 1  # -a 2: counter-driven move mix
    res = search.run(str(GOLD / "ref_tests" / phy), seed=509739986, algorithm=algorithm, batch=64, out=str(out),
                     max_seconds=60, verbose=False, device_proposals=on_device)
    assert res["best_length"] == score, res
    if topologies is not None:
        assert res["topologies"] == topologies
    lines = out.read_text().splitlines()
    assert len(lines) == res["topologies"] and all(l.startswith("(") and l.endswith(");") for l in lines)


@pytest.mark.parametrize("phy,score,chains,levels", [("test_treelength_6_thread_2.phy", 1628, 6, None),
                                                     ("test_treelength_7_thread_2.phy", 1006, 6, None),
                                                     ("test_treelength_7_thread_2.phy", 1006, 2, None),   # runs of 3 (the default for two chains)
                                                     ("test_treelength_6_thread_2.phy", 1628, 6, 3)])     # runs among six chains
def test_chains_on_one_gpu_reach_the_reference_optimum(tmp_path, phy, score, chains, levels):
    """`--chains R`: R restarts stepped together; the best over the chains is the reference's known optimum and the
    output holds each distinct best topology once - device-drawn steps, and with runs of accepted moves per scoring walk
    (`--run-levels`, host-drawn while a chain is hot)."""
    from lvb_amd import search
    out = tmp_path / "outtree"
    res = search.run_chains(str(GOLD / "ref_tests" / phy), seed=77, chains=chains, batch=64, out=str(out), max_seconds=60,
                            verbose=False, run_levels=levels)
    assert res["best_length"] == score and min(res["best_lengths"]) == score, res["best_lengths"]
    lines = out.read_text().splitlines()
    assert len(lines) == res["topologies"] >= 1 and len(set(lines)) == len(lines)
    assert all(l.startswith("(") and l.endswith(");") for l in lines)


def test_two_restarts_share_one_gpu_and_the_best_one_writes_the_trees(tmp_path):
    """The N-rank search flow (one independent restart per rank, min-reduce of the best length, the
    winner writes the output), rehearsed with two ranks on this box's single GPU over gloo - RCCL
    refuses two ranks on one device, and on a real node every rank has its own."""
    import os
    import socket
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "outtree"
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PYTHONPATH=str(root), LVBGPU_REHEARSE_ON_ONE_GPU="1")
        procs.append(subprocess.Popen([sys.executable, "-m", "lvb_amd.search", "-i",
                                       str(GOLD / "ref_tests" / "test_treelength_6_thread_2.phy"), "-s", "509739986",
                                       "-o", str(out), "--batch", "64", "--max-seconds", "60"],
                                      cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    texts = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-2000:]
        texts.append(o)
    printed = [t for t in texts if "Search Results" in t]
    assert len(printed) == 1                                # only the winning restart reports
    assert "Restarts (one per GPU):   2" in printed[0] and "Tree score:               1628" in printed[0]
    lines = out.read_text().splitlines()
    assert len(lines) == 1 and lines[0].startswith("(") and lines[0].endswith(");")


@pytest.mark.parametrize("alg,proposals", [(0, 0), (12, 0), (1, 2), (2, 1)])
def test_batched_search_at_the_bench_shape_under_every_kind_of_move(alg, proposals):
    """500 x 50 000 with NNI / TBR moves and batch sizes that grow to the re-root interval: the runs in which a stale
    length slot once surfaced.  The search must finish, and the tree it ends on must score, by a
    full evaluation on a second context, exactly what the search says."""
    from lvb_amd import api, host
    from tests import synth
    n, m = 500, 50000
    rows, minlen = host.prepare_alignment(synth.treelike_rows(n, m, 3))
    ctx = api.FitchContext(text_rows=rows)
    tree = host.HostTree(n, seed=5)
    start = tree.upload(ctx)
    p = host.anneal_defaults()
    p.seed = 5
    p.min_len_tree = minlen
    p.algorithm = alg
    p.device_proposals = proposals
    p.max_seconds = 4.0
    res, _ = host.anneal(ctx, tree, p)
    assert 0 < res["best_length"] < start
    _, left, right = tree.arrays()
    check = api.FitchContext(text_rows=rows)
    assert check.set_tree(left, right, tree.root) == res["final_length"] == ctx.current_length()
    check.close()
    tree.close()
    ctx.close()
