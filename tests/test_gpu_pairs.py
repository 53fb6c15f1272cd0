"""Two candidates per wave (fitch_walk_pair): the same lengths as one candidate per wave, whoever is paired with whom -
host-built batches (`LVBGPU_PAIR=n`: pairs from the full order of the programs read backwards), device-built ones (pairs
made by every generating workgroup among the sixteen candidates it has drawn), several chains in one launch (pairs never
cross a segment), odd counts (the last candidate walks alone), programs longer than one 64-token chunk (no sharing), and
batches below the threshold (not paired at all).  A context reads the switch when it is created."""
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
    """-> (a context that never pairs, one that pairs every batch of pair_min candidates and more)"""
    old = os.environ.get("LVBGPU_PAIR"), os.environ.get("LVBGPU_DIRECT_STEPS")
    try:
        os.environ["LVBGPU_PAIR"] = "0"
        plain = api.FitchContext(text_rows=rows)
        os.environ["LVBGPU_PAIR"] = str(pair_min)
        os.environ["LVBGPU_DIRECT_STEPS"] = "0"       # (direct steps - tiny batches straight to the host - are never paired)
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


def test_every_kind_of_draw_pairs_and_scores_the_same(mods):
    """NNI, SPR, TBR, mixed by position and alternating (the annealing schedule's) draws, one chain or several in a batch,
    one batch or two in flight: the paired walk's lengths are the plain walk's."""
    api, host = mods
    n, m = 200, 6000
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 41))
    never, paired = _contexts(api, rows, 1000)
    tree = host.HostTree(n, seed=7)
    assert tree.upload(never) == tree.upload(paired)
    walks = 0
    # kinds: 0 NNI, 1 SPR, 2 TBR, -1 by b % 3, -2 NNI / SPR alternating
    for B, kind in ((2500, 1), (4096, 2), (3000, -1), (1024, 0), (999, 0), (2048, -2)):
        draws = [(0, B, kind, 13 + B, B & 1, 0)]
        assert np.array_equal(paired.chains_propose_score(draws)[0], never.chains_propose_score(draws)[0]), (B, kind)
        walks += 1 if B >= 1000 else 0
        assert paired.paired_walks() == walks, (B, kind)
    for _ in range(100):
        e = tree.propose(1)
        never.commit(e)
        paired.commit(e)
        tree.apply(e)
    for c_ in (never, paired):
        c_.set_chains(2)
        for c in range(2):
            c_.select_chain(c)
            tree.upload(c_)
    for draws in ([(0, 900, 0, 5), (1, 601, 1, 6)], [(0, 17, 0, 7), (1, 1483, 2, 8)], [(0, 700, -2, 9, 1, 0), (1, 700, 0, 10)]):
        for g, w in zip(paired.chains_propose_score(draws), never.chains_propose_score(draws)):
            assert np.array_equal(g, w)
        walks += 1
        assert paired.paired_walks() == walks
    counts = paired.chains_submit(0, [(0, 2100, 0, 21)])
    paired.chains_submit(1, [(1, 2100, -2, 22)])
    a0, a1 = paired.chains_collect(0, counts)[0], paired.chains_collect(1, counts)[0]
    assert np.array_equal(a0, never.chains_propose_score([(0, 2100, 0, 21)])[0])
    assert np.array_equal(a1, never.chains_propose_score([(1, 2100, -2, 22)])[0])
    assert paired.paired_walks() == walks + 2 and never.paired_walks() == 0
    never.close()
    paired.close()


def test_a_workgroup_pairs_its_sixteen_by_their_longest_common_ends(mods):
    """Who walks with whom in a device-built batch: every candidate exactly once, partners from the same sixteen (and so
    from the same chain), odd runs end with a lone candidate - and for NNI neighbours, whose programs the host builder
    emits token for token as the device does, the pairs share exactly as many tokens as a greedy matching of each sixteen
    by the length of the programs' common end (longest first) does, more than half of what pairing the whole batch in
    the full order of the programs read backwards would share."""
    api, host = mods
    n, m = 120, 3000
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 43))
    never, paired = _contexts(api, rows, 1)
    never.close()
    tree = host.HostTree(n, seed=11)
    tree.upload(paired)
    for _ in range(40):
        e = tree.propose(1)
        paired.commit(e)
        tree.apply(e)

    def suffix(a, b):
        k = 0
        while k < len(a) and k < len(b) and a[len(a) - 1 - k] == b[len(b) - 1 - k]:
            k += 1
        return k

    for B, kind in ((1000, 0), (1000, 1), (333, 2), (47, -1)):
        paired.propose_score(B, kind, 3 * B + kind)
        pairs = paired.last_pairs(0)
        assert len(pairs) == (B + 1) // 2
        seen = np.zeros(B, int)
        for a, b in pairs:
            seen[a] += 1
            if b != 0xFFFFFFFF:
                seen[b] += 1
                assert a // 16 == b // 16
        assert (seen == 1).all() and int((pairs[:, 1] == 0xFFFFFFFF).sum()) == B % 2
        if kind != 0:
            continue
        progs = [tree.program(mode=0, edits=paired.proposal_edits(b)[0])["toks"] for b in range(B)]
        got = sum(suffix(progs[a], progs[b]) for a, b in pairs if b != 0xFFFFFFFF)
        want = 0
        for w0 in range(0, B, 16):
            idx = list(range(w0, min(B, w0 + 16)))
            sh = {(i, j): suffix(progs[i], progs[j]) for i in idx for j in idx if j > i}
            used = set()
            for (i, j), v in sorted(sh.items(), key=lambda kv: (-kv[1], -(16 * (kv[0][0] - w0) + kv[0][1] - w0))):   # (ties: the later pair, as the device)
                if i not in used and j not in used:
                    used |= {i, j}
                    want += v
        full = sorted(range(B), key=lambda i: tuple(progs[i][::-1].tolist()))
        best = sum(suffix(progs[full[i]], progs[full[i + 1]]) for i in range(0, B - 1, 2))
        assert got == want and 2 * got > best, (got, want, best)
    paired.close()
