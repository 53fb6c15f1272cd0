"""The measurement entry points behind bench.py's roofline line, and the batch life-cycle rules."""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    from lvb_amd import api, host
    assert api.device_count() >= 1
    rows, _ = host.prepare_alignment(synth.treelike_rows(60, 9000, 11))
    ctx = api.FitchContext(text_rows=rows)
    tree = host.HostTree(60, seed=12)
    tree.upload(ctx)
    yield api, host, ctx, tree
    ctx.close()


def test_walk_timing_counts_every_scoring_walk(setup):
    api, host, ctx, tree = setup
    ctx.walk_timing(True)
    for s in range(70):                      # more than the event ring holds: it drains on the way
        ctx.propose_score(64, 1, s)
    cands = [tree.propose(k % 3) for k in range(40)]
    ctx.score_batch(cands)
    ms, k = ctx.walk_timing_read()
    assert k == 71 and 0.0 < ms < 1000.0
    ctx.commit(tree.propose(1) if False else cands[0])    # commit walks are not counted
    tree.apply(cands[0])
    ms2, k2 = ctx.walk_timing_read()
    assert k2 == 71 and ms2 == ms
    ctx.walk_timing(False)
    ctx.walk_timing(4)                        # sampled: every 4th walk
    for s in range(9):
        ctx.propose_score(64, 1, s)
    assert ctx.walk_timing_read()[1] == 3     # walks 0, 4, 8
    ctx.walk_timing(False)
    ctx.propose_score(64, 1, 99)
    ctx.walk_timing(True)                     # enabling starts from zero
    assert ctx.walk_timing_read() == (0.0, 0)
    ctx.walk_timing(False)


def test_proposal_stats_equal_the_host_builders_counts(setup):
    api, host, ctx, tree = setup
    B = 500
    lens = ctx.propose_score(B, -1, 4242)
    st = ctx.proposal_stats()
    assert st["candidates"] == int((lens != np.iinfo(np.int64).max).sum())
    rows = comb = dirty = 0
    for b in range(B):
        if lens[b] == np.iinfo(np.int64).max:
            continue
        edits, _ = ctx.proposal_edits(b)
        prog = tree.program(mode=0, edits=edits)
        rows += len(prog["toks"])
        comb += len(prog["dsts"])
        dirty += prog["dirty"]
    assert (st["rows_read"], st["combines"], st["dirty_nodes"]) == (rows, comb, dirty)
    assert st["algorithmic_bytes"] == rows * ctx.nwords * 8


def test_probe_reads_at_a_plausible_rate(setup):
    api, host, ctx, tree = setup
    gbs = ctx.probe_l2(1024, 16, 5)
    assert 100.0 < gbs < 60000.0            # between a trickle and above any cache path of this chip
    with pytest.raises(api.LvbGpuError):
        ctx.probe_l2(0, 16, 5)


def test_lengths_before_launch_is_a_state_error_and_batches_may_outlive_their_context():
    from lvb_amd import api, host
    rows, _ = host.prepare_alignment(synth.treelike_rows(16, 300, 2))
    ctx = api.FitchContext(text_rows=rows)
    tree = host.HostTree(16, seed=3)
    tree.upload(ctx)
    b = ctx.build_batch([tree.propose(1) for _ in range(5)])
    with pytest.raises(api.LvbGpuError) as ei:
        b.lengths()
    assert ei.value.status == -5
    b.launch()
    assert (b.lengths() > 0).all()
    e = tree.propose(0)
    ctx.commit(e)                            # the batch's candidates were neighbours of the tree before this
    with pytest.raises(api.LvbGpuError) as ei:
        b.launch()
    assert ei.value.status == -5 and "changed" in str(ei.value)
    ctx.close()
    b.free()                                 # detached by lvbgpu_destroy: frees its own buffers only
