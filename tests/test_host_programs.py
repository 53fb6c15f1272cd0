"""Host logic without a GPU: the token programs the device will walk, checked by walking them in
numpy (tests/helpers.py) against the C oracle's getplen on the same topology + dirty flags."""
import numpy as np
import pytest

from tests import helpers, synth


@pytest.fixture(scope="module")
def host():
    from lvb_amd import host as h
    h.load_library()
    return h


@pytest.fixture(scope="module")
def ob():
    from oracle import binding
    return binding


def _oracle_tree(ob, enc, left, right, root):
    n, nwords = enc.shape
    ot = ob.OracleTree(n, nwords, enc)
    l64, r64 = np.asarray(left, dtype=np.int64), np.asarray(right, dtype=np.int64)
    ot.set_topology(helpers.parents_of(l64, r64), l64, r64, root)
    return ot


@pytest.mark.parametrize("n,m,seed", [(5, 7, 1), (9, 40, 2), (30, 200, 3), (64, 330, 4)])
def test_full_program_equals_oracle_getplen(host, ob, n, m, seed):
    enc = ob.encode_rows(synth.treelike_rows(n, m, seed))
    tree = host.HostTree(n, seed=seed)
    _, left, right = tree.arrays()
    ot = _oracle_tree(ob, enc, left, right, tree.root)
    expect = ot.getplen()
    prog = tree.program(mode=1)
    assert len(prog["toks"]) == n and len(prog["dsts"]) == n - 1  # D+3 rows, D+2 combines, D = n-3
    assert prog["dirty"] == n - 3
    # whole-tree programs only ever read leaf rows
    assert ((prog["toks"] & helpers.TOK_ROW_MASK) < n).all()
    total, produced, depth, _ = helpers.run_program(prog["toks"], prog["dsts"], enc)
    assert total == expect
    assert depth == prog["max_stack"] <= int(np.log2(n)) + 1   # Sethi-Ullman order keeps the stack shallow
    sets, ch = ot.all_sets(), ot.changes()
    assert sorted(produced) == list(range(n, 2 * n - 3))
    for node, (z, c) in produced.items():
        assert c == ch[node]
        assert np.array_equal(z, sets[node])


@pytest.mark.parametrize("n,m,seed", [(6, 20, 5), (12, 64, 6), (40, 300, 7)])
def test_candidate_programs_equal_oracle_incremental_getplen(host, ob, n, m, seed):
    enc = ob.encode_rows(synth.treelike_rows(n, m, seed))
    tree = host.HostTree(n, seed=seed)
    cur = None
    for step in range(60):
        _, left, right = tree.arrays()
        if cur is None:
            cur = _oracle_tree(ob, enc, left, right, tree.root)
            cur_len = cur.getplen()
        rows, ch = cur.all_sets(), cur.changes()
        s_all = int(ch[n:].sum())
        kind = step % 4
        new_root = -1
        if kind == 3:
            new_root = int((tree.root + 1 + step) % n)
            if new_root == tree.root:
                new_root = (new_root + 1) % n
            edits = tree.reroot_edits(new_root)
        else:
            edits = tree.propose(kind)
        prog = tree.program(mode=0, edits=edits, new_root=new_root)
        D = prog["dirty"]
        assert len(prog["toks"]) == D + 3 and len(prog["dsts"]) == D + 2
        assert list(prog["dsts"][-2:]) == [-1, -1]
        assert prog["max_stack"] <= 2, "NNI/SPR/TBR/re-root deltas fit the register stack"
        total, produced, depth, _ = helpers.run_program(prog["toks"], prog["dsts"], rows)
        assert depth == prog["max_stack"]
        got = s_all - int(sum(ch[d] for d in produced)) + total
        # the oracle on the candidate: same topology, the same nodes flagged dirty
        nl, nr = helpers.apply_edits(left, right, edits)
        cand = ob.OracleTree(n, enc.shape[1])
        cand.copy_from(cur)
        cand.set_topology(helpers.parents_of(nl, nr), nl, nr, new_root if new_root >= 0 else tree.root)
        cand.mark_dirty(sorted(produced))
        expect = cand.getplen()
        assert got == expect, f"step {step} kind {kind}"
        # ... and a from-scratch evaluation agrees (the dirty set was sufficient)
        lib = ob.load_oracle()
        assert expect == lib.lvbo_fitch_length_plain(n, enc.shape[1], enc, nl, nr, cand.root)
        csets, cch = cand.all_sets(), cand.changes()
        for node, (z, c) in produced.items():
            assert c == cch[node] and np.array_equal(z, csets[node])
        if step % 3 == 0:  # accept
            tree.apply(edits, new_root)
            cur, cur_len = cand, expect


def test_flagged_program_handles_arbitrary_dirty_flags(host, ob):
    """Strict compat: any subset of internal nodes may be flagged (sitestate[0]==0), including a
    dirty node under a clean parent; the reference recomputes exactly the flagged nodes."""
    n, m = 14, 96
    enc = ob.encode_rows(synth.treelike_rows(n, m, 9))
    tree = host.HostTree(n, seed=9)
    _, left, right = tree.arrays()
    rng = np.random.default_rng(9)
    for trial in range(40):
        cur = _oracle_tree(ob, enc, left, right, tree.root)
        cur.getplen()
        rows, ch = cur.all_sets(), cur.changes()
        flags = np.zeros(2 * n - 3, dtype=np.uint8)
        flags[n:] = rng.random(n - 3) < (0.15 + 0.02 * trial)
        # make the stored sets of flagged nodes garbage, as a real dirty node's would be
        prog = tree.program(mode=2, dirty=flags)
        total, produced, depth, root_ch = helpers.run_program(prog["toks"], prog["dsts"], rows)
        assert sorted(produced) == [int(i) for i in np.nonzero(flags)[0]]
        assert depth <= max(prog["max_stack"], 0) or depth == prog["max_stack"]
        base = int(ch[n:][flags[n:] == 0].sum())
        cur.mark_dirty(np.nonzero(flags)[0])
        expect = cur.getplen()
        assert base + total == expect
        csets, cch = cur.all_sets(), cur.changes()
        for node, (z, c) in produced.items():
            assert c == cch[node] and np.array_equal(z, csets[node])


def test_bad_edits_are_rejected(host):
    from lvb_amd import api
    tree = host.HostTree(8, seed=3)
    _, left, right = tree.arrays()
    u = 8  # an internal node
    bad = np.zeros(1, dtype=api.EDIT_DTYPE)
    bad[0] = (u, u, left[u])  # a node as its own child
    with pytest.raises(api.LvbGpuError):
        tree.program(mode=0, edits=bad)
    bad[0] = (u, -1, left[u])  # exactly one child
    with pytest.raises(api.LvbGpuError):
        tree.program(mode=0, edits=bad)
    bad[0] = (99, 1, 2)  # out of range
    with pytest.raises(api.LvbGpuError):
        tree.program(mode=0, edits=bad)


def test_random_topologies_are_valid_and_varied(host):
    seen = set()
    for seed in range(1, 40):
        t = host.HostTree(12, seed=seed)
        p, l, r = t.arrays()
        assert t.root == 0 and p[0] == -1
        assert (l[1:12] == -1).all() and (r[1:12] == -1).all()
        assert (l[12:] >= 0).all() and (r[12:] >= 0).all()
        seen.add(helpers.splits_of(l, r, 0, 12))
    assert len(seen) > 30


def test_topology_hash_identifies_unrooted_topologies(host):
    """The treestack mirror: same bipartition set <=> same hash, whatever the root or the moves
    that led there; different topologies differ."""
    n = 14
    tree = host.HostTree(n, seed=21)
    seen, canon = {}, {}
    rng = np.random.default_rng(5)
    for step in range(400):
        _, l, r = tree.arrays()
        key = helpers.splits_of(l, r, tree.root, n)
        h = tree.topology_hash()
        assert seen.setdefault(key, h) == h
        # the exact identity behind the hash: one canonical form per bipartition set, whatever the root ...
        c = tree.canonical()
        assert canon.setdefault(key, c) == c and len(c) == 2 * n - 3
        assert sorted(x for x in c if x >= 0) == list(range(1, n)) and c.count(-1) == n - 2
        # ... and whatever the numbering of the internal nodes
        perm = np.arange(2 * n - 3)
        perm[n:] = n + rng.permutation(n - 3)
        l2, r2 = np.full(2 * n - 3, -1, np.int32), np.full(2 * n - 3, -1, np.int32)
        for v in range(2 * n - 3):
            if l[v] >= 0:
                l2[perm[v]], r2[perm[v]] = perm[r[v]], perm[l[v]]          # children swapped as well
        assert host.HostTree(left=l2, right=r2, root=tree.root).canonical() == c
        if step % 5 == 4:
            new_root = (tree.root + 3) % n
            tree.apply(tree.reroot_edits(new_root), new_root)
            _, l2, r2 = tree.arrays()
            assert helpers.splits_of(l2, r2, tree.root, n) == key     # re-rooting keeps the topology
            assert tree.topology_hash() == h
        else:
            tree.apply(tree.propose(step % 3))
    assert len(set(seen.values())) == len(seen) > 100
    assert len(set(canon.values())) == len(canon) == len(seen)      # different topologies, different forms

