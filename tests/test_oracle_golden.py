"""The CPU oracle (oracle/fitch_oracle.c) against the committed golden vectors, which hold what
the real reference's getplen/DNAToBinary/matchange produced (tests/golden/gen_golden.py).
Runs anywhere (no GPU, no /root/reference)."""
import numpy as np
import pytest

from tests import goldenlib, helpers


@pytest.fixture(scope="module")
def ob():
    from oracle import binding
    binding.load_oracle()
    return binding


def _tree_for(ob, g, enc, case):
    ot = ob.OracleTree(g.n, g.nwords, enc)
    l, r = case["left"].astype(np.int64), case["right"].astype(np.int64)
    ot.set_topology(helpers.parents_of(l, r), l, r, int(case["root"]))
    return ot


@pytest.mark.parametrize("name", goldenlib.names())
def test_oracle_reproduces_reference_vectors(ob, name):
    g = goldenlib.Golden(name)
    enc = g.enc(ob.encode_rows)                       # pins lvbo_encode_row + the column cut
    rows = g.rows()
    assert len(rows[0]) == g.m
    assert ob.load_oracle().lvbo_words_per_row(g.m) == g.nwords
    if "text" in g.z:
        assert ob.min_tree_length(rows) == g.min_len_tree
    clean = {}                                        # case index -> OracleTree in its post-getplen state
    step = 1 if g.nwords < 1000 else 3
    for k in range(g.cases):
        c = g.case(k)
        base = int(c["base"])
        if base >= 0 and base not in clean:
            # the current tree this case was mutated from: a full evaluation of its topology
            bt = _tree_for(ob, g, enc, g.case(base))
            bt.getplen()
            clean[base] = bt
        if k % step and base >= 0 and g.nwords >= 1000:
            continue
        ot = ob.OracleTree(g.n, g.nwords, enc)
        if base >= 0:
            ot.copy_from(clean[base])
        l, r = c["left"].astype(np.int64), c["right"].astype(np.int64)
        ot.set_topology(helpers.parents_of(l, r), l, r, int(c["root"]))
        ot.mark_dirty(np.nonzero(c["dirty"])[0])
        assert ot.getplen() == int(c["length"]), f"{name} case {k}"
        assert np.array_equal(ot.changes()[g.n:], c["changes"][g.n:])
        sets = ot.all_sets()
        assert goldenlib.crc(sets) == int(c["sets_crc"])
        if "sets" in c:
            assert np.array_equal(sets, c["sets"])
        if int(c["threads_length"]) >= 0:
            assert int(c["threads_length"]) == int(c["length"])  # the reference's OpenMP branch agreed
        clean = {b: t for b, t in clean.items() if b >= k - 40}  # bound memory


@pytest.mark.parametrize("name", ["ref_test_treelength_6_thread_2", "ref_stock_100x1000", "edge_m33"])
def test_sliced_walk_equals_serial_walk(ob, name):
    """The reference's OpenMP site-slice arithmetic (TreeEvaluation.c:64-181), restated."""
    g = goldenlib.Golden(name)
    enc = g.enc(ob.encode_rows)
    for k in range(0, g.cases, 5):
        c = g.case(k)
        if int(c["base"]) >= 0:
            continue
        for nslices in (2, 3):
            ot = _tree_for(ob, g, enc, c)
            ot.mark_all_dirty()
            assert ot.getplen_sliced(nslices, max(g.nwords // nslices, 1)) == int(c["length"])
            assert np.array_equal(ot.changes()[g.n:], c["changes"][g.n:])
            assert goldenlib.crc(ot.all_sets()) == int(c["sets_crc"])


def test_exhaustive_small_alignments_reach_reference_optimum(ob):
    """The reference's black-box tests pin the OPTIMUM (test_treelength_{1,2}, test_matrix_phylip_
    length: scores 1, 5, 9).  For 5 taxa all 15 unrooted topologies can be scored."""
    import itertools
    import json
    from pathlib import Path
    manifest = json.loads((Path(__file__).parent / "golden" / "ref_tests.json").read_text())
    want = {c["infile"]: c["expect"]["Tree score"] for c in manifest["cases"] if "Tree score" in c["expect"]}
    lib = ob.load_oracle()
    for stem in ("test_treelength_1", "test_treelength_2", "test_matrix_phylip_length"):
        g = goldenlib.Golden(f"ref_{stem}")
        assert g.n == 5
        enc = g.enc(ob.encode_rows)
        best = None
        # 5 taxa: root 0, internal nodes 5 and 6; all ways to hang 4 leaves
        for a, b in itertools.combinations(range(1, 5), 2):
            c, d = [x for x in range(1, 5) if x not in (a, b)]
            shapes = [
                ((5, 6), {5: (a, b), 6: (c, d)}),           # ((a,b),(c,d))
                ((c, 5), {5: (d, 6), 6: (a, b)}),           # (c,(d,(a,b)))
                ((d, 5), {5: (c, 6), 6: (a, b)}),           # (d,(c,(a,b)))
            ]
            for top, inner in shapes:
                left = np.full(7, -1, dtype=np.int64)
                right = np.full(7, -1, dtype=np.int64)
                left[0], right[0] = top
                for v, (x, y) in inner.items():
                    left[v], right[v] = x, y
                score = lib.lvbo_fitch_length_plain(5, g.nwords, enc, left, right, 0)
                ot = ob.OracleTree(5, g.nwords, enc)
                ot.set_topology(helpers.parents_of(left, right), left, right, 0)
                assert ot.getplen() == score
                best = score if best is None else min(best, score)
        assert best == want[f"{stem}.phy"], stem
