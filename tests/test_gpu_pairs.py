"""Two candidates per wave (fitch_walk_pair, `LVBGPU_PAIR=n`): the same lengths as one candidate per wave, whoever is
paired with whom - host-built batches (pairs from the full order of the programs read backwards), device-built ones
(pairs from the generator's keys, sorted by pair_kernel), several chains in one launch (pairs never cross a segment),
odd counts (the last candidate walks alone), programs longer than one 64-token chunk (walked alone), and batches below
the threshold (not paired at all).  A context reads the switch when it is created."""
import os

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from lvb_amd import api, host
    assert api.device_count() >= 1
    return api, host


def _contexts(api, rows, pair_min):
    plain = api.FitchContext(text_rows=rows)
    old = os.environ.get("LVBGPU_PAIR"), os.environ.get("LVBGPU_DIRECT_STEPS")
    os.environ["LVBGPU_PAIR"] = str(pair_min)
    os.environ["LVBGPU_DIRECT_STEPS"] = "0"       # (direct steps - tiny batches straight to the host - are never paired)
    try:
        paired = api.FitchContext(text_rows=rows)
    finally:
        for key, val in zip(("LVBGPU_PAIR", "LVBGPU_DIRECT_STEPS"), old):
            if val is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = val
    return plain, paired


@pytest.mark.parametrize("n,m,walk", [(9, 100, 0), (60, 3000, 10), (500, 50000, 75)])
def test_paired_walk_gives_the_unpaired_lengths(mods, n, m, walk):
    api, host = mods
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 23))
    plain, paired = _contexts(api, rows, 1)
    tree = host.HostTree(n, seed=29)
    for _ in range(walk):
        tree.apply(tree.propose(1))
    assert tree.upload(plain) == tree.upload(paired)
    rng = np.random.default_rng(5)
    for B in (1, 2, 3, 64, 257, 1500 if n >= 60 else 300):
        cands = [tree.propose(int(rng.integers(0, 3))) for _ in range(B)]
        cands += cands[: B // 3]                                                     # identical programs: the whole walk shared
        want = plain.score_batch(cands)
        assert np.array_equal(paired.score_batch(cands), want), B
        for kind in (0, 1, 2, -1):
            seed = 100 * B + kind
            assert np.array_equal(paired.propose_score(B, kind, seed), plain.propose_score(B, kind, seed)), (B, kind)
    # accept on both and go on: the paired context's commits are ordinary commit walks
    lens = plain.propose_score(512, 1, 77)
    assert np.array_equal(paired.propose_score(512, 1, 77), lens)
    b = int(np.argmin(lens))
    edits, _ = plain.proposal_edits(b)
    assert paired.commit(paired.proposal_edits(b)[0]) == plain.commit(edits) == lens[b]
    assert np.array_equal(paired.propose_score(511, -1, 78), plain.propose_score(511, -1, 78))
    assert plain.paired_walks() == 0 and paired.paired_walks() >= 6 * 5 + 2           # every scoring walk above was a paired one
    plain.close()
    paired.close()


def test_pairs_stay_inside_their_chain(mods):
    """Several resident trees in one launch: a segment per chain, each sorted and paired on its own."""
    api, host = mods
    n, m, R = 40, 2500, 5
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 31))
    plain, paired = _contexts(api, rows, 1)
    trees = [host.HostTree(n, seed=100 + c) for c in range(R)]
    for ctx in (plain, paired):
        ctx.set_chains(R)
        for c, t in enumerate(trees):
            ctx.select_chain(c)
            t.upload(ctx)
    draws = [(c, 1 + 37 * c, (-1, 0, 1, 2, -1)[c], 900 + c) for c in range(R)]       # (chain, count, kind, seed): odd and tiny counts too
    a, b = plain.chains_propose_score(draws), paired.chains_propose_score(draws)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and paired.paired_walks() == 1
    # each chain's lengths are what the chain gives alone
    for c in range(R):
        plain.select_chain(c)
        assert np.array_equal(plain.propose_score(draws[c][1], draws[c][2], draws[c][3]), b[c])
    plain.close()
    paired.close()


def test_long_programs_walk_alone_and_small_batches_are_left_alone(mods):
    """A caterpillar's root-ward paths give programs of more than 64 tokens: never paired, walked chunk by chunk by the
    same kernel; with a threshold above the batch nothing is paired (same entry points, plain kernel)."""
    api, host = mods
    n, m = 150, 1200
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 37))
    left, right = np.full(2 * n - 3, -1, np.int32), np.full(2 * n - 3, -1, np.int32)
    left[0], right[0] = 1, n                       # root leaf 0 holds (1, n); node n + i holds (leaf i + 2, n + i + 1)
    for i in range(n - 3):
        v = n + i
        left[v] = i + 2
        right[v] = v + 1 if i < n - 4 else n - 1
    tree = host.HostTree(left=left, right=right, root=0, seed=3)
    plain, paired = _contexts(api, rows, 1)
    assert tree.upload(plain) == tree.upload(paired)
    for kind in (1, 2, -1):
        lens = plain.propose_score(901, kind, 5 + kind)
        assert np.array_equal(paired.propose_score(901, kind, 5 + kind), lens)
    st = paired.proposal_stats()
    assert st["rows_read"] > 64 * st["candidates"] // 2            # deep: many programs are longer than a chunk
    cands = [paired.proposal_edits(b)[0] for b in range(0, 901, 3)]
    assert np.array_equal(paired.score_batch(cands), lens[::3])
    plain.close()
    paired.close()
    plain, thresh = _contexts(api, rows, 600)
    tree.upload(plain), tree.upload(thresh)
    for B, walks in ((599, 0), (600, 1), (601, 2)):
        assert np.array_equal(thresh.propose_score(B, 1, B), plain.propose_score(B, 1, B))
        assert thresh.paired_walks() == walks
    plain.close()
    thresh.close()


def test_long_programs_are_paired_when_the_library_is_left_to_choose(mods):
    """LVBGPU_PAIR=auto: the library walks two candidates per wave where the loads saved are many - device-built batches of
    2048 candidates and more whose programs are long (estimated from the tree's mean node depth) - and nowhere else; the
    order is made by the last workgroups of the generator's own launch.  Same lengths as the plain walk either way."""
    api, host = mods
    n, m = 200, 6000
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 41))
    other, never = _contexts(api, rows, 0)            # LVBGPU_PAIR=0 ...
    other.close()
    old = os.environ.get("LVBGPU_PAIR")
    os.environ["LVBGPU_PAIR"] = "auto"
    try:
        auto = api.FitchContext(text_rows=rows)       # ... and left to the library
    finally:
        if old is None:
            os.environ.pop("LVBGPU_PAIR", None)
        else:
            os.environ["LVBGPU_PAIR"] = old
    tree = host.HostTree(n, seed=7)
    assert tree.upload(never) == tree.upload(auto)
    # a fresh random tree is shallow: short programs, nothing is paired
    assert np.array_equal(auto.propose_score(2500, 1, 11), never.propose_score(2500, 1, 11))
    assert auto.paired_walks() == 0
    # mixed by accepted moves its paths get long: big batches are paired, small ones never
    for _ in range(600):
        e = tree.propose(1)
        never.commit(e)
        auto.commit(e)
        tree.apply(e)
    for B, kind in ((2048, 1), (4096, 2), (3000, -1)):
        assert np.array_equal(auto.propose_score(B, kind, 13 + B), never.propose_score(B, kind, 13 + B)), B
    assert auto.paired_walks() == 3 and never.paired_walks() == 0
    assert np.array_equal(auto.propose_score(2047, 1, 5), never.propose_score(2047, 1, 5)) and auto.paired_walks() == 3
    # two batches in flight, both paired
    counts = auto.chains_submit(0, [(0, 2100, 1, 21)])
    auto.chains_submit(1, [(0, 2100, 1, 22)])
    a0, a1 = auto.chains_collect(0, counts)[0], auto.chains_collect(1, counts)[0]
    assert np.array_equal(a0, never.propose_score(2100, 1, 21)) and np.array_equal(a1, never.propose_score(2100, 1, 22))
    assert auto.paired_walks() == 5
    never.close()
    auto.close()
