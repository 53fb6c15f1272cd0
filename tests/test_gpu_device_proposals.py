"""Neighbourhoods drawn on the device (lvbgpu_propose_score): every candidate must be a move our
host generators - themselves checked against the reference's mutate_* - can reproduce from the
reported parameters, with identical edits, and its device-built program must give the length the
host-built program gives."""
import numpy as np
import pytest

from tests import helpers, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from lvb_amd import api, host
    assert api.device_count() >= 1
    return api, host


def _replay(tree, info, api):
    kind, a, b, c = (int(x) for x in info)
    if kind == 0:
        return tree.nni_edits(a, bool(b))
    if kind == 1 or c < 0:
        return tree.spr_edits(a, b)
    return tree.tbr_edits(a, b, c)


@pytest.mark.parametrize("n,m,B", [(7, 40, 256), (60, 3000, 768), (500, 50000, 1536)])
def test_device_moves_replay_on_host_and_score_the_same(mods, n, m, B):
    api, host = mods
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 41))
    ctx = api.FitchContext(text_rows=rows)
    tree = host.HostTree(n, seed=43)
    tree.upload(ctx)
    for round_ in range(3):
        for kind in (0, 1, 2, -1):
            seed = 1000 * round_ + 17 * (kind + 2)
            lens = ctx.propose_score(B, kind, seed)
            assert np.array_equal(lens, ctx.propose_score(B, kind, seed))      # a function of (seed, b)
            assert not np.array_equal(lens, ctx.propose_score(B, kind, seed + 1))
            ctx.propose_score(B, kind, seed)                                    # make it the current batch again
            assert (lens < np.iinfo(np.int64).max).all()                        # nothing overflowed at these sizes
            step = 1 if B <= 256 else 7
            cands, picked, kinds_seen = [], [], set()
            for b in range(0, B, step):
                edits, info = ctx.proposal_edits(b)
                assert info[0] == (kind if kind >= 0 else b % 3)
                assert helpers.edit_key(edits) == helpers.edit_key(_replay(tree, info, api)), (kind, b, info)
                cands.append(edits)
                picked.append(b)
                kinds_seen.add((int(info[0]), int(info[3]) >= 0))
            assert np.array_equal(ctx.score_batch(cands), lens[picked])        # host-built programs agree
            if kind == 2 and n >= 60:
                assert (2, True) in kinds_seen                                  # real TBR re-rootings occurred
        # accept something and go on from the new tree
        b = int(np.argmin(lens))
        edits, _ = ctx.proposal_edits(b)
        assert ctx.commit(edits) == lens[b]
        with pytest.raises(api.LvbGpuError):                                    # the batch is stale now
            ctx.proposal_edits(b)
        tree.apply(edits)
    ctx.close()


def test_device_draws_cover_the_neighbourhood(mods):
    """With enough draws every NNI of a small tree shows up, and SPR sources/destinations spread."""
    api, host = mods
    n = 9
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, 64, 5))
    ctx = api.FitchContext(text_rows=rows)
    tree = host.HostTree(n, seed=6)
    tree.upload(ctx)
    B = 2048
    ctx.propose_score(B, 0, 99)
    seen = {tuple(int(x) for x in ctx.proposal_edits(b)[1][:3]) for b in range(B)}
    assert len(seen) == 2 * (n - 3)                                             # every (u, side)
    ctx.propose_score(B, 1, 99)
    pairs = {tuple(int(x) for x in ctx.proposal_edits(b)[1][1:3]) for b in range(0, B, 2)}
    assert len(pairs) > 40
    ctx.close()


def test_device_draws_follow_the_uniform_law(mods):
    """The admissible sets are the reference's and the device draws uniformly over them (chi-square at n = 20):
    NNI over the 2 (n - 3) (node, side) pairs; SPR sources over the 2n - 6 nodes that may be pruned, and, given the most
    frequent source, destinations over ITS admissible set.  (The reference's own randpint gives the two END values of
    its range half weight - it rounds uni() * upper to nearest, RandomNumberGenerator.c:243-246 - so its draws are not
    exactly uniform; the device's are.  Exact-trajectory runs draw on the host with the reference's generator.)"""
    api, host = mods
    n = 20
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, 200, 5))
    ctx = api.FitchContext(text_rows=rows)
    tree = host.HostTree(n, seed=11)
    tree.upload(ctx)
    _, left, right, root = ctx.topology()
    parent = np.full(2 * n - 3, -1)
    for v in range(2 * n - 3):
        if left[v] >= 0:
            parent[left[v]] = parent[right[v]] = v

    def chi2(counts, expected):
        return float(((counts - expected) ** 2 / expected).sum())

    def loose(dof):            # far out in the tail (p ~ 1e-6): the seeds are fixed, this guards against a skewed generator
        return dof + 5.0 * (2.0 * dof) ** 0.5

    B = 6800
    ctx.propose_score(B, 0, 1234)
    info = np.array([ctx.proposal_edits(b)[1] for b in range(B)])
    cells = (info[:, 1] - n) * 2 + info[:, 2]                     # (u, which child of u was given away)
    counts = np.bincount(cells, minlength=2 * (n - 3)).astype(float)
    assert (counts > 0).all() and chi2(counts, B / (2 * (n - 3))) < loose(2 * (n - 3) - 1)

    B = 12000
    ctx.propose_score(B, 1, 4321)
    info = np.array([ctx.proposal_edits(b)[1] for b in range(B)])
    src, dest = info[:, 1], info[:, 2]
    ok_src = [v for v in range(2 * n - 3) if v != root and v != left[root] and v != right[root]]
    counts = np.array([(src == v).sum() for v in ok_src], dtype=float)
    assert counts.sum() == B and chi2(counts, B / len(ok_src)) < loose(len(ok_src) - 1)
    s0 = ok_src[int(np.argmax(counts))]
    sp = parent[s0]
    ss = right[sp] if left[sp] == s0 else left[sp]

    def below(v, a):
        while v != -1:
            if v == a:
                return True
            v = parent[v]
        return False
    ok_dest = [v for v in range(2 * n - 3) if v not in (s0, sp, ss, root) and not below(v, s0)]
    got = dest[src == s0]
    counts = np.array([(got == v).sum() for v in ok_dest], dtype=float)
    assert counts.sum() == len(got)                                # nothing outside the admissible set
    assert chi2(counts, len(got) / len(ok_dest)) < loose(len(ok_dest) - 1)
    ctx.close()


def test_move_schedules_of_the_reference(mods):
    """-a 0: NNI/SPR alternate by parity; -a 1: kinds drawn with the given probabilities."""
    api, host = mods
    n = 40
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, 500, 8))
    ctx = api.FitchContext(text_rows=rows)
    tree = host.HostTree(n, seed=9)
    tree.upload(ctx)
    B = 1200
    for parity in (0, 1):
        ctx.propose_score_mixed(B, 0.0, 0.0, parity, 5)
        kinds = [int(ctx.proposal_edits(b)[1][0]) for b in range(0, B, 3)]
        assert kinds == [1 if (parity + b) & 1 else 0 for b in range(0, B, 3)]
    ctx.propose_score_mixed(B, 0.2, 0.3, -1, 6)
    kinds = np.array([int(ctx.proposal_edits(b)[1][0]) for b in range(B)])
    frac = [(kinds == k).mean() for k in (0, 1, 2)]
    assert abs(frac[0] - 0.2) < 0.05 and abs(frac[1] - 0.3) < 0.05 and abs(frac[2] - 0.5) < 0.05
    ctx.propose_score_mixed(B, 0.0, 0.0, -1, 7)                                  # all TBR (t = t0)
    assert all(int(ctx.proposal_edits(b)[1][0]) == 2 for b in range(0, B, 5))
    with pytest.raises(api.LvbGpuError):
        ctx.propose_score_mixed(B, 0.8, 0.5, -1, 7)
    ctx.close()


def _as_map(edits):
    return {int(e["node"]): (int(e["left"]), int(e["right"])) for e in edits}


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_moves_named_by_the_host_score_and_rewrite_exactly_like_host_generators(kind):
    """lvbgpu_score_moves: the device turns (kind, a, b, c) into rewrites + program.  Lengths must equal
    lvbgpu_score_batch on the host generators' edits for the same moves, and the rewrites the device
    reports must be those edits child for child, side for side (a search that reproduces the reference's
    trajectory indexes nodes and sides by number)."""
    from lvb_amd import api, host
    n, m, B = 60, 700, 512
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 21))
    ctx = api.FitchContext(text_rows=rows)
    tree = host.HostTree(n, seed=5)
    tree.upload(ctx)
    rng = host.RefRng(1234)
    try:
        for round_ in range(3):
            # draw moves with the reference's generators on the host: parameters, and their edits
            moves = [tree.ref_draw_move(rng, kind) for _ in range(B)]
            cands = [tree.move_edits(mv) for mv in moves]
            got = ctx.score_moves(moves)
            want = ctx.score_batch(cands)
            assert np.array_equal(got, want)
            for b in (0, 1, B // 2, B - 1):
                dev_edits, info = ctx.proposal_edits(b)
                assert _as_map(dev_edits) == _as_map(cands[b]), (moves[b], info)
            # accept one through the device's own rewrites and go on from the new tree
            dev_edits, _ = ctx.proposal_edits(7)
            assert ctx.commit(dev_edits) == got[7]
            tree.apply(dev_edits)
    finally:
        ctx.close()


def test_moves_the_generators_could_not_make_are_refused():
    from lvb_amd import api, host
    n = 12
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, 64, 2))
    ctx = api.FitchContext(text_rows=rows)
    tree = host.HostTree(n, seed=2)
    tree.upload(ctx)
    p, l, r = tree.arrays()
    root = tree.root
    internal = next(v for v in range(n, 2 * n - 3) if p[v] != root and l[v] >= n)
    try:
        bad = [(0, 3, 0, -1),                      # NNI at a leaf
               (1, root, internal, -1),            # prune the root
               (1, int(l[root]), internal, -1),    # prune a child of the root
               (1, internal, internal, -1),        # graft on itself
               (1, internal, int(l[internal]), -1),  # graft inside the pruned subtree
               (2, internal, root, -1),            # graft on the root
               (2, internal, int(p[internal]), 0),  # graft on its own parent
               (7, 0, 0, 0)]
        for mv in bad:
            with pytest.raises(api.LvbGpuError) as ei:
                ctx.score_moves([mv])
            assert ei.value.status == -6, mv
    finally:
        ctx.close()


def test_batches_that_grow_and_shrink_keep_their_length_slots_clean(mods):
    """The recycled batches zero their length slots off the critical path and remember that they did.  A
    buffer that grows must forget it - its new memory holds whatever was there before, and the allocator likes
    to hand back the old address (a search at 500 taxa died with a candidate scored -4.6e18 that way).  Sizes
    that cross every growth step of the buffers, each batch against the host-built programs of the same moves."""
    api, host = mods
    n, m = 40, 600
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 77))
    ctx = api.FitchContext(text_rows=rows)
    tree = host.HostTree(n, seed=78)
    tree.upload(ctx)
    rng = host.RefRng(79)
    for B in (8, 513, 64, 700, 1100, 100, 2600, 300, 5300, 11000, 640):
        moves = np.array([tree.ref_draw_move(rng, b % 3) for b in range(B)], dtype=api.MOVE_DTYPE)
        edits = [tree.move_edits(mv) for mv in moves]
        want = ctx.score_batch(edits)
        assert want.min() > 0
        assert np.array_equal(ctx.score_moves(moves), want), B
        assert np.array_equal(ctx.score_batch(edits), want), B
        lens = ctx.propose_score(B, -1, 5 * B)
        assert lens.min() > 0 and np.array_equal(lens, ctx.propose_score(B, -1, 5 * B)), B
    ctx.close()


def test_device_moves_on_a_caterpillar(mods):
    """The deepest tree there is (every internal node has a leaf child): root-ward paths of more than 64 nodes (several
    lane-parallel trips per path run), TBR re-rootings along the whole spine, destinations mostly inside the pruned
    subtree (long rejection runs).  Every device move replays on the host and scores what the host-built program scores;
    accepted moves keep the device's own tables (rebuilt on the device) in step."""
    api, host = mods
    n, m = 150, 2100
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 19))
    left, right = np.full(2 * n - 3, -1, np.int32), np.full(2 * n - 3, -1, np.int32)
    # root leaf 0 holds (1, n); internal node n + i holds (leaf i + 2, n + i + 1); the last one two leaves
    left[0], right[0] = 1, n
    for i in range(n - 3):
        v = n + i
        left[v] = i + 2
        right[v] = v + 1 if i < n - 4 else n - 1
    ctx = api.FitchContext(text_rows=rows)
    tree = host.HostTree(left=left, right=right, root=0, seed=3)
    tree.upload(ctx)
    for round_ in range(4):
        B = 600
        lens = ctx.propose_score(B, -1, 40 + round_)
        assert (lens < np.iinfo(np.int64).max).all()
        cands = []
        for b in range(0, B, 5):
            edits, info = ctx.proposal_edits(b)
            assert helpers.edit_key(edits) == helpers.edit_key(_replay(tree, info, api)), (round_, b, info)
            cands.append(edits)
        assert np.array_equal(ctx.score_batch(cands), lens[::5])
        # accept through the multi-chain path (device-side table rebuild), then draw again from the new tree
        draws = ctx.chains_propose_score([(0, 64, -1, 900 + round_)])[0]
        b = int(np.argmin(draws))
        ctx.chains_commit([(0, b)])
        assert ctx.current_length() == draws[b]
        _, l, r, root = ctx.topology()
        tree = host.HostTree(left=l, right=r, root=root, seed=4 + round_)
    ctx.close()


def test_lone_steps_without_the_watcher_take_the_copy_path():
    """A lone step's lengths normally come through the walk's watcher waves (fitch_walk<.., 2>); LVBGPU_WATCHER=0 - and
    launches of more tile groups than the slots' arrival counts can hold - read them back with a copy instead.  The
    replay test once more in a child process with the watcher off keeps that path covered."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LVBGPU_WATCHER="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_device_proposals.py"), "-m", "gpu",
                        "-q", "-x", "-k", "replay_on_host or grow_and_shrink", "-p", "no:cacheprovider"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout, r.stdout[-500:]
