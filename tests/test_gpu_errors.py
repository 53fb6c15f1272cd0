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


def test_a_wait_gives_up_at_the_contexts_limit(mods):
    """no host spin without a deadline.  The stream is kept busy by a clock-watching kernel for 400 ms
    (lvbgpu_debug_stall), the wait limit is 50 ms: collecting a batch queued behind it returns LVBGPU_E_HIP naming what
    was waited for, within the limit's order of magnitude - and the context is serviceable again once the stream has
    drained."""
    import time
    api, host = mods
    rows, _ = host.prepare_alignment(synth.treelike_rows(30, 900, 5))
    ctx = api.FitchContext(text_rows=rows)
    tree = host.HostTree(30, seed=6)
    tree.upload(ctx)
    want = ctx.propose_score(64, -1, 11)
    for _ in range(6):   # (the first uses of the recycled step batches allocate pinned buffers, which waits for the device)
        ctx.score_batch([tree.propose(1) for _ in range(9)])
    ctx.set_wait_limit(0.05)
    for collect_kind in ("watcher", "score_batch"):
        ctx._chk(ctx.lib.lvbgpu_debug_stall(ctx.h, 400))
        t0 = time.perf_counter()
        with pytest.raises(api.LvbGpuError) as ei:
            if collect_kind == "watcher":
                ctx.propose_score(64, -1, 11)                # device-built batch: its lengths come through the watcher waves
            else:
                ctx.score_batch([tree.propose(1) for _ in range(9)])
        took = time.perf_counter() - t0
        assert ei.value.status == -3 and ("wait limit" in str(ei.value)), str(ei.value)
        assert 0.04 < took < 0.35, (collect_kind, took)                     # gave up at the limit, not when the stall ended
        ctx.synchronize()                                     # the stall kernel is bounded: the stream drains
    ctx.set_wait_limit(30.0)
    assert np.array_equal(ctx.propose_score(64, -1, 11), want)   # nothing resident was harmed
    with pytest.raises(api.LvbGpuError):
        ctx.set_wait_limit(0.0)
    ctx.close()


def test_watcher_waves_that_wait_in_vain_give_up_and_the_host_says_so(mods, monkeypatch):
    """The walk's lengths reach the host through watcher waves at the end of its grid, which wait for every candidate's
    waves to have counted themselves in.  That wait is bounded like every other; what happens when it runs out is otherwise
    never seen: LVBGPU_DEBUG_STARVE_WATCHER (read when a context is created) makes the watchers wait for one wave more than
    there are and look only a few thousand times.  They give up, the host gets LVBGPU_E_STATE naming the watcher within
    milliseconds - nothing hangs, the resident tree is untouched - and a context created without the switch is served."""
    import time
    api, host = mods
    rows, _ = host.prepare_alignment(synth.treelike_rows(30, 900, 5))
    monkeypatch.setenv("LVBGPU_DEBUG_STARVE_WATCHER", "1")
    starved = api.FitchContext(text_rows=rows)
    monkeypatch.delenv("LVBGPU_DEBUG_STARVE_WATCHER")
    plain = api.FitchContext(text_rows=rows)
    tree = host.HostTree(30, seed=6)
    length = tree.upload(plain)
    assert tree.upload(starved) == length
    want = plain.propose_score(300, -1, 11)
    for B in (300, 64, 2000):
        t0 = time.perf_counter()
        with pytest.raises(api.LvbGpuError) as ei:
            starved.propose_score(B, -1, 11)
        assert ei.value.status == -5 and "watcher" in str(ei.value), str(ei.value)
        assert time.perf_counter() - t0 < 2.0
    # what does not go through the watchers still works on that context, and its tree is what it was
    assert starved.current_length() == length
    cands = [tree.propose(1) for _ in range(9)]
    assert np.array_equal(starved.score_batch(cands), plain.score_batch(cands))
    assert np.array_equal(plain.propose_score(300, -1, 11), want)
    starved.close()
    plain.close()
