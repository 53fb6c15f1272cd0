"""The pieces of the reference-trajectory search (lvb_amd/csrc/refsearch.cpp, refrng.hpp) against the
compiled reference (oracle/_ref): the random stream value for value, the random start tree and every
proposal / re-root record for record - including which side each child sits on and how many draws
each step consumed, since one draw out of step derails a whole run."""
import numpy as np
import pytest

from tests import helpers, synth


@pytest.fixture(scope="module")
def ob():
    from oracle import binding
    if binding.load_ref() is None:
        pytest.skip("oracle/_ref/liblvbref.so not available")
    return binding


@pytest.fixture(scope="module")
def host():
    from lvb_amd import host as h
    h.load_library()
    return h


@pytest.mark.parametrize("seed", [0, 1, 12345, 509739986, 465380177, 900000000])
def test_random_stream_matches_value_for_value(ob, host, seed):
    rr = ob.RefRun(rows=synth.treelike_rows(6, 16, 1), seed=seed)
    try:
        rr.reseed(seed)
        g = host.RefRng(seed)
        for k in range(3000):
            if k % 3 == 0:
                assert g.uni() == rr.uni()  # bit for bit
            else:
                upper = (k * 7919) % 1000 if k % 2 else (k * 104729) % (1 << 40)
                assert g.randpint(upper) == rr.randpint(upper)
    finally:
        rr.close()


def test_seed_range_is_the_reference_s(host):
    from lvb_amd import api
    for bad in (-1, 900000001):
        with pytest.raises(api.LvbGpuError):
            host.RefRng(bad)


@pytest.mark.parametrize("n", [5, 6, 9, 33, 100])
def test_random_start_tree_record_for_record(ob, host, n):
    rr = ob.RefRun(rows=synth.treelike_rows(n, 32, n), seed=1)
    try:
        for seed in (1, 77, 4242, 12345):
            rr.reseed(seed)
            g = host.RefRng(seed)
            for _ in range(3):  # successive trees from one stream (GetSoln draws two)
                rr.random_tree()
                _, rl, rrr, _, _ = rr.tree(0)
                t = g.random_tree(n)
                _, l, r = t.arrays()
                assert t.root == 0 and rr.root(0) == 0
                assert np.array_equal(l, rl) and np.array_equal(r, rrr)
                assert g.uni() == rr.uni()  # both streams consumed the same number of draws
    finally:
        rr.close()


@pytest.mark.parametrize("n,steps", [(5, 1500), (6, 1500), (12, 3000), (40, 2500), (100, 1200)])
def test_proposals_and_reroots_record_for_record(ob, host, n, steps):
    seed = 1000 + n
    rr = ob.RefRun(rows=synth.treelike_rows(n, 48, n), seed=seed)
    try:
        rr.reseed(seed)
        g = host.RefRng(seed)
        rr.random_tree()
        tree = g.random_tree(n)
        rr.getplen(0)
        for s in range(steps):
            if s % 37 == 36:
                e, nr = tree.ref_arbreroot(g)
                assert rr.arbreroot() == nr
                tree.apply(e, nr)
                rp, rl, rrr, _, _ = rr.tree(0)
                p, l, r = tree.arrays()
                assert np.array_equal(l, rl) and np.array_equal(r, rrr) and np.array_equal(p, rp), f"re-root at step {s}"
                rr.getplen(0)
                continue
            kind = s % 3
            rr.mutate(kind)
            e = tree.ref_propose(g, kind)
            rp, rl, rrr, _, _ = rr.tree(1)
            _, l, r = tree.arrays()
            nl, nr_ = helpers.apply_edits(l, r, e)
            assert np.array_equal(nl, rl) and np.array_equal(nr_, rrr), f"kind {kind} at step {s}"
            if (s * 2654435761) % 5 < 2:  # accept some (Solve.c:323 SwapTrees)
                rr.swap()
                tree.apply(e)
                p, _, _ = tree.arrays()
                assert np.array_equal(p, rp)
                rr.getplen(0)
        assert g.uni() == rr.uni()
    finally:
        rr.close()
