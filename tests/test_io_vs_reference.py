"""Alignment reader and tree writer of the host mirror against the reference's own
(MSAInput.cpp read_phylip, TreeOperations.c ur_print), on the reference's test inputs."""
from pathlib import Path

import numpy as np
import pytest

GOLD = Path(__file__).resolve().parent / "golden" / "ref_tests"


@pytest.fixture(scope="module")
def ob():
    from oracle import binding
    if binding.load_ref() is None:
        pytest.skip("oracle/_ref/liblvbref.so not available")
    return binding


@pytest.mark.parametrize("phy", sorted(p.name for p in GOLD.glob("*.phy")))
def test_phylip_reader_and_cut_match_reference(ob, phy):
    from lvb_amd import host
    names, rows = host.read_phylip(GOLD / phy)
    rr = ob.RefRun(path=str(GOLD / phy), fmt=0, seed=1)
    try:
        assert len(rows) == rr.n and len(rows[0]) == rr.original_m
        assert [n.rstrip() for n in names] == [t.rstrip() for t in rr.titles()]
        cut, min_len = host.prepare_alignment(rows)
        assert cut == rr.rows()
        assert min_len == rr.min_len
    finally:
        rr.close()


def test_newick_matches_reference_treeprint(ob):
    from lvb_amd import host
    rr = ob.RefRun(path=str(GOLD / "test_treelength_6_thread_2.phy"), fmt=0, seed=5)
    try:
        names = rr.titles()
        for step in range(12):
            if step % 4 == 3:
                rr.getplen(0)
                rr.arbreroot()
            elif step:
                rr.mutate(step % 3)
                rr.swap()
            _, l, r, _, _ = rr.tree(0)
            t = host.HostTree(left=l.astype(np.int32), right=r.astype(np.int32), root=rr.root(0))
            assert host.newick(t, names) == rr.treeprint(0)
    finally:
        rr.close()


def test_reader_rejects_ragged_file(tmp_path):
    from lvb_amd import host
    bad = tmp_path / "bad.phy"
    bad.write_text(" 3 4\nName_1    ACGT\nName_2    ACG\nName_3    ACGT\n")
    with pytest.raises(ValueError) as ei:
        host.read_phylip(bad)
    assert "different length" in str(ei.value)
