"""Error behaviour of the C ABI on a real device: status codes instead of exits, state untouched
by rejected calls, and the reference's `changes > 0` assertion as a status."""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from lvb_amd import api, host
    assert api.device_count() >= 1
    return api, host


def test_call_order_and_bad_arguments(mods):
    api, host = mods
    rows, _ = host.prepare_alignment(synth.treelike_rows(12, 200, 3))
    ctx = api.FitchContext(text_rows=rows)
    tree = host.HostTree(12, seed=4)
    e = tree.propose(1)
    for call in (lambda: ctx.score_batch([e]), lambda: ctx.commit(e), lambda: ctx.current_length(),
                 lambda: ctx.propose_score(8, 1, 1), lambda: ctx.proposal_edits(0), lambda: ctx.changes()):
        with pytest.raises(api.LvbGpuError) as ei:
            call()
        assert ei.value.status == -5 and "first" in str(ei.value)                    # LVBGPU_E_STATE
    _, left, right = tree.arrays()
    bad = left.copy()
    bad[12] = bad[13]                                                                # a child claimed twice
    with pytest.raises(api.LvbGpuError) as ei:
        ctx.set_tree(bad, right, 0)
    assert ei.value.status == -6
    with pytest.raises(api.LvbGpuError) as ei:
        ctx.set_tree(left, right, 13)                                                # root must be a leaf
    assert ei.value.status == -6
    with pytest.raises(api.LvbGpuError):
        api.FitchContext(text_rows=rows, device=99)
    ctx.close()


def test_rejected_edits_leave_the_resident_tree_alone(mods):
    api, host = mods
    rows, _ = host.prepare_alignment(synth.treelike_rows(20, 600, 5))
    ctx = api.FitchContext(text_rows=rows)
    tree = host.HostTree(20, seed=6)
    length = tree.upload(ctx)
    before = (ctx.topology(), ctx.changes().copy())
    u = 25
    bads = [np.array([(u, u, 3)], dtype=api.EDIT_DTYPE),                              # own child
            np.array([(u, -1, 3)], dtype=api.EDIT_DTYPE),                             # one child only
            np.array([(99, 1, 2)], dtype=api.EDIT_DTYPE),                             # out of range
            np.array([(u, 1, 2)], dtype=api.EDIT_DTYPE)]                              # leaves a cycle / orphaned nodes
    good = tree.propose(1)
    for bad in bads:
        with pytest.raises(api.LvbGpuError) as ei:
            ctx.score_batch([good, bad])
        assert ei.value.status == -6 and "candidate 1" in str(ei.value)
        with pytest.raises(api.LvbGpuError):
            ctx.commit(bad)
    assert ctx.current_length() == length
    after = (ctx.topology(), ctx.changes())
    assert all(np.array_equal(a, b) for a, b in zip(before[0][:3], after[0][:3])) and before[0][3] == after[0][3]
    assert np.array_equal(before[1], after[1])
    assert ctx.score_batch([good])[0] > 0                                             # still serviceable
    ctx.close()


def test_zero_length_is_the_reference_assertion(mods):
    """An alignment of nothing but N scores 0: the reference asserts changes > 0
    (TreeEvaluation.c:267); the ABI reports LVBGPU_E_ZEROLEN instead of exiting."""
    api, host = mods
    enc = np.full((6, 2), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
    ctx = api.FitchContext(enc)
    tree = host.HostTree(6, seed=2)
    _, left, right = tree.arrays()
    with pytest.raises(api.LvbGpuError) as ei:
        ctx.set_tree(left, right, tree.root)
    assert ei.value.status == -8 and "changes > 0" in str(ei.value)
    ctx.close()
