"""Size-independent properties at BASELINE.json's full shapes, where running the CPU reference for
every case would take minutes: the device must agree WITH ITSELF across independent routes to the
same number, and with the reference on a thin sample.

  P1  incremental score of a candidate == full evaluation of the candidate's topology
  P2  score_batch == commit (the two kernels' variants) and commit leaves changes summing to the length
  P3  re-rooting never changes the length (Fitch length is root-independent)
  P4  undoing a move restores length, per-node changes and every node set (crc)
  P5  lengths >= MinimumTreeLength and <= (n-1) * m
  P6  a batch scored twice, or split in two, gives identical lengths (atomics commute)
"""
import numpy as np
import pytest

from tests import goldenlib, helpers, synth

pytestmark = pytest.mark.gpu

# (name, taxa, sites, move kinds, batch)   cfg2, cfg3, cfg5 of BASELINE.json
CONFIGS = [
    ("cfg2_64x10k_nni", 64, 10000, (0,), 1024),
    ("cfg3_500x50k_spr", 500, 50000, (1,), 1024),
    ("cfg3_500x50k_mixed", 500, 50000, (0, 1, 2), 512),
    ("cfg5_2000x200k_tbr", 2000, 200000, (2,), 96),
]


@pytest.fixture(scope="module")
def mods():
    from lvb_amd import api, host
    assert api.device_count() >= 1
    return api, host


def _inverse(left, right, edits):
    """Edits that put back what `edits` replaced."""
    from lvb_amd import api
    inv = np.zeros(len(edits), dtype=api.EDIT_DTYPE)
    for i, e in enumerate(edits):
        inv[i] = (e["node"], left[e["node"]], right[e["node"]])
    return inv


@pytest.mark.parametrize("name,n,m,kinds,B", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_full_size_properties(mods, name, n, m, kinds, B):
    api, host = mods
    rows, min_len = host.prepare_alignment(synth.treelike_rows(n, m, 17))
    m_cut = len(rows[0])
    ctx = api.FitchContext(text_rows=rows)
    del rows
    tree = host.HostTree(n, seed=23)
    length = tree.upload(ctx)
    assert min_len <= length <= (n - 1) * m_cut                                     # P5
    for _ in range(30):                                                             # leave the start shape
        e = tree.propose(kinds[0])
        length = ctx.commit(e)
        tree.apply(e)

    _, left, right = tree.arrays()
    root = tree.root
    cands = [tree.propose(kinds[b % len(kinds)]) for b in range(B)]
    lens = ctx.score_batch(cands)
    assert (lens >= min_len).all() and (lens <= (n - 1) * m_cut).all()              # P5
    assert np.array_equal(lens, ctx.score_batch(cands))                             # P6 (repeat)
    half = B // 2
    assert np.array_equal(lens, np.concatenate([ctx.score_batch(cands[:half]), ctx.score_batch(cands[half:])]))

    # P1: a sample of candidates re-evaluated from scratch as whole trees (full-batch kernel path)
    sample = list(range(0, B, max(1, B // 12)))[:12]
    lefts, rights = [], []
    for b in sample:
        nl, nr = helpers.apply_edits(left, right, cands[b])
        lefts.append(nl.astype(np.int32))
        rights.append(nr.astype(np.int32))
    full = ctx.score_full_batch(np.stack(lefts), np.stack(rights), [root] * len(sample))
    assert np.array_equal(full, lens[sample])

    # P3: re-root as a candidate and as a commit
    new_root = (root + 7) % n
    rr_edits = tree.reroot_edits(new_root)
    assert ctx.score_batch([rr_edits], roots=[new_root])[0] == length

    # P2 + P4: commit a candidate, check bookkeeping, undo it, everything is back
    before_changes = ctx.changes()
    probe_nodes = [n, n + (n - 3) // 2, 2 * n - 4]
    before_sets = [goldenlib.crc(ctx.sets(v)) for v in probe_nodes]
    b = sample[len(sample) // 2]
    got = ctx.commit(cands[b])
    assert got == lens[b] == ctx.current_length()                                   # P2
    ch = ctx.changes()
    # internal changes + the two root combines = length: the root part is what is left over
    assert 0 <= got - int(ch[n:].sum()) <= 2 * m_cut
    back = ctx.commit(_inverse(left, right, cands[b]))
    assert back == length                                                           # P4
    assert np.array_equal(ctx.changes(), before_changes)
    assert [goldenlib.crc(ctx.sets(v)) for v in probe_nodes] == before_sets
    ctx.close()


def test_reference_sample_at_cfg3(mods):
    """A thin slice of cfg3 against the real reference (the rest of cfg3 parity is in the golden
    vectors synth_500x50000 and the properties above)."""
    from oracle import binding as ob
    if ob.load_ref() is None:
        pytest.skip("oracle/_ref/liblvbref.so did not travel")
    api, host = mods
    n, m = 500, 50000
    rows = synth.treelike_rows(n, m, 29)
    rr = ob.RefRun(rows=rows, seed=31)
    try:
        ctx = api.FitchContext(text_rows=rr.rows())
        _, cl, cr, _, _ = rr.tree(0)
        assert ctx.set_tree(cl.astype(np.int32), cr.astype(np.int32), rr.root(0)) == rr.getplen(0)
        cands, expect = [], []
        for b in range(24):
            rr.mutate(b % 3)
            _, nl, nr, _, _ = rr.tree(1)
            cands.append(api.edits_between(cl, cr, nl, nr))
            expect.append(rr.getplen(1))
        assert np.array_equal(ctx.score_batch(cands), np.array(expect))
        ctx.close()
    finally:
        rr.close()


def test_reference_sample_at_cfg5(mods):
    """A thin slice of cfg5 (2000 x 200 000, TBR: the HBM-resident size, 401 MB of state sets) against the real
    reference: the full evaluation of its start tree, eight TBR neighbours scored from edits, then one accepted
    and its per-node changes compared.  (Further cfg5 parity: golden synth_2000x200000 and the properties above.)"""
    from oracle import binding as ob
    if ob.load_ref() is None:
        pytest.skip("oracle/_ref/liblvbref.so did not travel")
    api, host = mods
    n, m = 2000, 200000
    rr = ob.RefRun(rows=synth.treelike_rows(n, m, 37), seed=41)
    try:
        ctx = api.FitchContext(text_rows=rr.rows())
        _, cl, cr, _, _ = rr.tree(0)
        assert ctx.set_tree(cl.astype(np.int32), cr.astype(np.int32), rr.root(0)) == rr.getplen(0)
        assert np.array_equal(ctx.changes()[n:], rr.tree(0)[3][n:])
        cands, expect = [], []
        for b in range(8):
            rr.mutate(2)
            _, nl, nr, _, _ = rr.tree(1)
            cands.append(api.edits_between(cl, cr, nl, nr))
            expect.append(rr.getplen(1))
        assert np.array_equal(ctx.score_batch(cands), np.array(expect))
        # the last one accepted on both sides (SwapTrees / lvbgpu_commit): same bookkeeping afterwards
        rr.swap()
        assert ctx.commit(cands[-1]) == expect[-1]
        assert np.array_equal(ctx.changes()[n:], rr.tree(0)[3][n:])
        for v in (n, n + 777, 2 * n - 4):
            assert np.array_equal(ctx.sets(v), rr.sets(0, v))
        ctx.close()
    finally:
        rr.close()
