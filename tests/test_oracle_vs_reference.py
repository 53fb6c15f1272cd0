"""oracle/fitch_oracle.c pinned directly against the compiled reference (oracle/_ref), on random
alignments, trees and mutation walks, running the oracle's getplen IN PLACE on the reference's own
tree blocks.  Skipped where the reference is not available; test_oracle_golden.py then carries the pin."""
import ctypes as C

import numpy as np
import pytest

from tests import synth


@pytest.fixture(scope="module")
def ob():
    from oracle import binding
    if binding.load_ref() is None:
        pytest.skip("oracle/_ref/liblvbref.so not available")
    return binding


def test_struct_layouts_the_adapter_relies_on(ob):
    lay = ob.layout()
    assert lay["node_size"] == 40 and C.sizeof(ob.Node) == 40
    assert (lay["off_parent"], lay["off_left"], lay["off_right"], lay["off_changes"], lay["off_sitestate"]) == \
        (0, 8, 16, 24, 32)
    assert lay["msa_size"] == 120 and lay["params_size"] == 4040
    assert (lay["off_nthreads"], lay["off_slice"], lay["off_m"], lay["off_n"], lay["off_nbranches"],
            lay["off_bytes"], lay["off_nwords"]) == (0, 4, 8, 24, 40, 48, 72)


@pytest.mark.parametrize("n,m,seed,gen", [(5, 3, 1, "u"), (11, 47, 2, "i"), (40, 700, 3, "t"), (90, 3000, 4, "t")])
def test_oracle_getplen_in_place_on_reference_blocks(ob, n, m, seed, gen):
    rows = {"u": synth.uniform_rows, "i": synth.iupac_rows, "t": synth.treelike_rows}[gen](n, m, seed)
    if gen == "i":
        rows[0] = b"A" * m
        rows[1] = b"C" * m
    ra = ob.RefRun(rows=rows, seed=seed)   # scored by the reference
    rb = ob.RefRun(rows=rows, seed=seed)   # same trees, scored by the oracle in place
    lib = ob.load_oracle()
    todo = np.zeros(max(ra.nbranches - ra.n, 1), dtype=np.int64)
    try:
        assert np.array_equal(ra.enc(), ob.encode_rows(ra.rows()))
        assert ob.cut_constant_columns(rows) == ra.rows()
        assert ob.min_tree_length(ra.rows()) == ra.min_len
        assert lib.lvbo_tree_bytes(ra.nbranches, ra.nwords) == ra.lib.refh_tree_bytes(ra.h)

        def oracle_getplen(which):
            return lib.lvbo_getplen(rb.tree_block(which), rb.n, rb.nbranches, rb.nwords, rb.root(which), todo)

        assert ra.getplen(0) == oracle_getplen(0)
        for step in range(90):
            kind = step % 3
            ra.reseed(100 + step)
            ra.mutate(kind)
            rb.reseed(100 + step)
            rb.mutate(kind)
            assert ra.getplen(1) == oracle_getplen(1), f"step {step}"
            ta, tb = ra.tree(1), rb.tree(1)
            assert np.array_equal(ta[3][n:], tb[3][n:])
            assert np.array_equal(ra.all_sets(1), rb.all_sets(1))
            if step % 4 == 3:
                ra.swap()
                rb.swap()
            if step % 20 == 19:
                ra.reseed(900 + step)
                ra.arbreroot()
                rb.reseed(900 + step)
                rb.arbreroot()
                assert ra.getplen(0) == oracle_getplen(0)
    finally:
        ra.close()
        rb.close()


def test_word_step_matches_reference_on_random_words(ob):
    """lvbo_combine vs the reference's inline-asm popcnt step, via 3-taxon-wide probes: a tree of
    5 taxa whose first internal node combines two chosen rows."""
    rng = np.random.default_rng(5)
    lib = ob.load_oracle()
    for _ in range(200):
        x = int(rng.integers(0, 2**63, dtype=np.uint64)) | 0x1111111111111111
        y = int(rng.integers(0, 2**63, dtype=np.uint64)) | 0x2222222222222222
        ch = C.c_long(0)
        z = lib.lvbo_combine(x, y, C.byref(ch))
        # per nibble, by definition
        zz, cc = 0, 0
        for k in range(16):
            a, b = (x >> 4 * k) & 15, (y >> 4 * k) & 15
            if a & b:
                zz |= (a & b) << 4 * k
            else:
                zz |= (a | b) << 4 * k
                cc += 1
        assert z == zz and ch.value == cc
