"""Property-based checks of the host program builder (no GPU): for ANY tree, ANY sequence of moves
and ANY dirty-flag pattern the program must (1) be well formed, (2) evaluate - walked in numpy - to
the oracle's getplen on the same inputs."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from tests import helpers, synth


@pytest.fixture(scope="module")
def host():
    from lvb_amd import host as h
    h.load_library()
    return h


@pytest.fixture(scope="module")
def ob():
    from oracle import binding
    binding.load_oracle()
    return binding


def _well_formed(prog, n):
    toks, dsts = prog["toks"], prog["dsts"]
    assert toks[0] & helpers.TOK_FRESH and not (toks[0] & helpers.TOK_PUSH)
    fresh = int(((toks & helpers.TOK_FRESH) != 0).sum())
    pushes = int(((toks & helpers.TOK_PUSH) != 0).sum())
    merges = int(((toks >> helpers.TOK_MERGE_SHIFT) & helpers.TOK_MERGE_MASK).sum())
    assert len(dsts) == len(toks) - fresh + merges          # one dst per combine
    produced = [int(d) for d in dsts if d >= 0]
    assert len(produced) == len(set(produced))              # every dirty node produced once
    assert all(n <= d < 2 * n - 3 for d in produced)
    return fresh, pushes, merges


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(n=st.integers(5, 48), seed=st.integers(1, 10**6), walk=st.lists(st.integers(0, 3), min_size=0, max_size=12),
       flag_p=st.floats(0.0, 1.0))
def test_any_tree_any_flags(host, ob, n, seed, walk, flag_p):
    m = 64
    enc = ob.encode_rows(synth.iupac_rows(n, m, seed))
    tree = host.HostTree(n, seed=seed)
    for k in walk:                                           # move away from the start shape
        if k == 3:
            nr = (tree.root + 1 + seed) % n
            if nr != tree.root:
                tree.apply(tree.reroot_edits(nr), nr)
        else:
            tree.apply(tree.propose(k))
    _, left, right = tree.arrays()
    l64, r64 = left.astype(np.int64), right.astype(np.int64)
    cur = ob.OracleTree(n, enc.shape[1], enc)
    cur.set_topology(helpers.parents_of(l64, r64), l64, r64, tree.root)
    full_len = cur.getplen()

    # whole-tree program
    prog = tree.program(mode=1)
    fresh, pushes, merges = _well_formed(prog, n)
    assert pushes == merges                                  # every saved sibling set is merged back
    total, produced, depth, _ = helpers.run_program(prog["toks"], prog["dsts"], enc)
    assert total == full_len and depth == prog["max_stack"]

    # arbitrary dirty flags (strict compat semantics)
    rng = np.random.default_rng(seed)
    flags = np.zeros(2 * n - 3, dtype=np.uint8)
    flags[n:] = rng.random(n - 3) < flag_p
    rows, ch = cur.all_sets(), cur.changes()
    prog = tree.program(mode=2, dirty=flags)
    _well_formed(prog, n)
    total, produced, _, _ = helpers.run_program(prog["toks"], prog["dsts"], rows)
    assert sorted(produced) == [int(i) for i in np.nonzero(flags)[0]]
    cur.mark_dirty(np.nonzero(flags)[0])
    assert int(ch[n:][flags[n:] == 0].sum()) + total == cur.getplen()

    # a candidate: edits -> dirty set -> program
    for kind in (0, 1, 2):
        edits = tree.propose(kind)
        prog = tree.program(mode=0, edits=edits)
        fresh, pushes, merges = _well_formed(prog, n)
        assert pushes == merges <= 1 and prog["max_stack"] <= 1 + (kind == 2)
        nl, nr = helpers.apply_edits(left, right, edits)
        lib = ob.load_oracle()
        want = lib.lvbo_fitch_length_plain(n, enc.shape[1], enc, nl, nr, tree.root)
        rows2, ch2 = cur.all_sets(), cur.changes()           # cur is clean again after getplen above
        total, produced, _, _ = helpers.run_program(prog["toks"], prog["dsts"], rows2)
        assert int(ch2[n:].sum()) - int(sum(ch2[d] for d in produced)) + total == want
