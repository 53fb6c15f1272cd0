"""Several resident trees (chains) in one context: each slot must behave exactly like a context of its own."""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from lvb_amd import api, host
    assert api.device_count() >= 1
    return api, host


@pytest.mark.parametrize("n,m", [(9, 70), (60, 5000)])
def test_every_chain_slot_equals_a_context_of_its_own(mods, n, m):
    api, host = mods
    R = 3
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 51))
    multi = api.FitchContext(text_rows=rows)
    multi.set_chains(R)
    singles = [api.FitchContext(text_rows=rows) for _ in range(R)]
    trees = [host.HostTree(n, seed=60 + c) for c in range(R)]
    for c in range(R):
        multi.select_chain(c)
        assert trees[c].upload(multi) == trees[c].upload(singles[c])
    rng = np.random.default_rng(3)
    for step in range(12):
        for c in rng.permutation(R):
            c = int(c)
            multi.select_chain(c)
            one = singles[c]
            assert multi.current_length() == one.current_length()
            cands = [trees[c].propose(k % 3) for k in range(17)]
            want = one.score_batch(cands)
            assert np.array_equal(multi.score_batch(cands), want)
            # neighbourhoods drawn on the device: same tree, same seed => same moves, same lengths
            seed = 100 * step + c
            lens = multi.propose_score(40, -1, seed)
            assert np.array_equal(lens, one.propose_score(40, -1, seed))
            b = int(np.argmin(lens))
            e_multi, info_m = multi.proposal_edits(b)
            e_one, info_o = one.proposal_edits(b)
            assert np.array_equal(e_multi, e_one) and np.array_equal(info_m, info_o)
            # accept on both sides: lengths, per-node changes and node sets stay equal
            pick = cands[int(np.argmin(want))] if step % 2 else e_multi
            assert multi.commit(pick) == one.commit(pick)
            trees[c].apply(pick)
            assert np.array_equal(multi.changes(), one.changes())
            for v in (n, n + (n - 3) // 2, 2 * n - 4):
                assert np.array_equal(multi.sets(v), one.sets(v))
            if step % 5 == 4:   # a re-root as a commit
                nr = (trees[c].root + 2) % n
                ed = trees[c].reroot_edits(nr)
                assert multi.commit(ed, root=nr) == one.commit(ed, root=nr)
                trees[c].apply(ed, nr)
    # the other chains were never disturbed by what happened to one of them
    for c in range(R):
        multi.select_chain(c)
        assert multi.current_length() == singles[c].current_length()
        assert np.array_equal(multi.all_sets(), singles[c].all_sets())
        p1, l1, r1, root1 = multi.topology()
        p2, l2, r2, root2 = singles[c].topology()
        assert np.array_equal(l1, l2) and np.array_equal(r1, r2) and root1 == root2
    for s in singles:
        s.close()
    multi.close()


def test_chain_calls_reject_bad_arguments(mods):
    api, host = mods
    rows, _ = host.prepare_alignment(synth.treelike_rows(8, 40, 2))
    ctx = api.FitchContext(text_rows=rows)
    for bad in (0, 65):
        with pytest.raises(api.LvbGpuError):
            ctx.set_chains(bad)
    ctx.set_chains(2)
    with pytest.raises(api.LvbGpuError):
        ctx.select_chain(2)
    ctx.select_chain(1)
    with pytest.raises(api.LvbGpuError) as ei:      # no tree in this slot yet
        ctx.current_length()
    assert ei.value.status == -5
    ctx.close()


@pytest.mark.parametrize("n,m,R", [(12, 300, 4), (500, 50000, 6)])
def test_one_step_over_all_chains_equals_the_chains_stepped_one_by_one(mods, n, m, R):
    """lvbgpu_chains_propose_score / lvbgpu_chains_commit: one generator launch, one walk and one commit walk for
    all chains must give each chain exactly what the single-tree calls give it (same seeds => same moves, same
    lengths; same picks => same trees, node sets and per-node changes)."""
    api, host = mods
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 71))
    multi = api.FitchContext(text_rows=rows)
    multi.set_chains(R)
    ref = api.FitchContext(text_rows=rows)          # stepped chain by chain through the single-tree calls
    ref.set_chains(R)
    for c in range(R):
        t = host.HostTree(n, seed=80 + c)
        for ctx in (multi, ref):
            ctx.select_chain(c)
            t.upload(ctx)
    rng = np.random.default_rng(9)
    kinds = [0, 1, 2, -1]
    for step in range(10):
        active = sorted(rng.choice(R, size=int(rng.integers(1, R + 1)), replace=False).tolist())
        draws = [(c, int(rng.integers(1, 70 if n > 100 else 30)), kinds[(step + c) % 4], 1000 * step + c) for c in active]
        got = multi.chains_propose_score(draws)
        picks = []
        for (c, count, kind, seed), lens in zip(draws, got):
            ref.select_chain(c)
            want = ref.propose_score(count, kind, seed)
            assert np.array_equal(lens, want), (step, c)
            ok = np.nonzero(lens < np.iinfo(np.int64).max)[0]
            if len(ok) and (step + c) % 3 != 2:      # some chains accept, some do not
                b = int(ok[np.argmin(lens[ok])])
                picks.append((c, b))
                edits, _ = ref.proposal_edits(b)
                assert ref.commit(edits) == lens[b]
        if picks:
            multi.chains_commit(picks)
        for c in range(R):
            multi.select_chain(c)
            ref.select_chain(c)
            assert multi.current_length() == ref.current_length(), (step, c)
            assert np.array_equal(multi.changes(), ref.changes())
            pm, pr = multi.topology(), ref.topology()
            assert all(np.array_equal(a, b) for a, b in zip(pm[:3], pr[:3])) and pm[3] == pr[3]
            for v in (n, n + (n - 3) // 2, 2 * n - 4):
                assert np.array_equal(multi.sets(v), ref.sets(v))
    # stale picks and bad arguments are refused
    got = multi.chains_propose_score([(0, 5, 1, 1), (1, 5, 1, 2)])
    multi.chains_commit([(0, 0)])
    with pytest.raises(api.LvbGpuError) as ei:
        multi.chains_commit([(0, 1)])                  # chain 0's tree has changed since the draw
    assert ei.value.status == -5
    multi.chains_commit([(1, 4)])                      # chain 1's draw is still good
    for bad in ([(2, 0)], [(1, 9)], [(0, 0), (0, 1)]):
        with pytest.raises(api.LvbGpuError):
            multi.chains_commit(bad)
    with pytest.raises(api.LvbGpuError):
        multi.chains_propose_score([(0, 5, 1, 1), (0, 5, 1, 2)])
    multi.close()
    ref.close()


def _run_chains(api, host, rows, min_len, n, seeds, max_proposals, algorithm, run_levels=0, lanes=1):
    ctx = api.FitchContext(text_rows=rows)
    trees = [host.HostTree(n, seed=1000 + s) for s in seeds]
    params = []
    for s in seeds:
        p = host.anneal_defaults()
        p.seed = 7000 + s
        p.algorithm = algorithm
        p.batch = 256
        p.t0 = 0.0                      # every chain estimates its own starting temperature
        p.min_len_tree = min_len
        p.max_proposals = max_proposals
        p.log_cap = 64
        p.run_levels = run_levels
        p.lanes = lanes
        params.append(p)
    res, log = host.anneal_chains(ctx, trees, params)
    final = []
    for c, t in enumerate(trees):
        _, l, r = t.arrays()
        if lanes == 1 or c < len(seeds) // lanes:        # (lane 0 is the caller's context; the other lanes' are gone)
            ctx.select_chain(c)
            # the host's mirror is the library's tree, and the resident length is what the chain believes
            _, ll, lr, lroot = ctx.topology()
            assert np.array_equal(l, ll) and np.array_equal(r, lr) and t.root == lroot
            assert ctx.current_length() == res[c]["final_length"]
        final.append((l.copy(), r.copy(), t.root, t.best_count()))
    ctx.close()
    return res, final, log


@pytest.mark.parametrize("algorithm", [0, 1])
def test_a_chains_trajectory_does_not_depend_on_how_many_chains_run_beside_it(mods, algorithm):
    """R = 1 vs R = 6: identical decisions per chain for fixed seeds (the chains share launches, nothing else)."""
    api, host = mods
    n, m = 40, 1500
    rows, min_len = host.prepare_alignment(synth.treelike_rows(n, m, 91))
    seeds = [3, 4, 5, 6, 7, 8]
    many, many_final, log = _run_chains(api, host, rows, min_len, n, seeds, 2500, algorithm)
    keys = ("start_length", "best_length", "final_length", "consumed", "accepted", "temperatures", "device_steps", "scored",
            "reroots", "topologies", "t_final")
    for pick in (0, 3, 5):
        one, one_final, _ = _run_chains(api, host, rows, min_len, n, [seeds[pick]], 2500, algorithm)
        assert {k: one[0][k] for k in keys} == {k: many[pick][k] for k in keys}
        assert all(np.array_equal(a, b) for a, b in zip(one_final[0][:2], many_final[pick][:2]))
        assert one_final[0][2:] == many_final[pick][2:]
    assert all(r["consumed"] == 2500 and r["best_length"] <= r["start_length"] for r in many)
    assert [b for _, b in log] == sorted((b for _, b in log), reverse=True)      # the shared log only ever improves
    assert log[-1][1] == min(r["best_length"] for r in many)


def test_lanes_leave_every_chains_trajectory_alone(mods):
    """lvbhost_anneal_params::lanes on the HIP scorer: the chains dealt to two and three contexts (lvbgpu_fork) whose steps
    overlap on the device, served by one host thread in the order their lengths arrive.  Every chain's run is what one
    lane gives it (CPU tier: tests/test_anneal_chains_cpu.py)."""
    api, host = mods
    n, m = 40, 1500
    rows, min_len = host.prepare_alignment(synth.treelike_rows(n, m, 91))
    seeds = [3, 4, 5, 6, 7, 8, 9]
    keys = ("start_length", "best_length", "final_length", "consumed", "accepted", "temperatures", "device_steps", "scored",
            "reroots", "topologies", "t_final")
    one, one_final, _ = _run_chains(api, host, rows, min_len, n, seeds, 2500, 1)
    for lanes in (2, 3):
        many, many_final, log = _run_chains(api, host, rows, min_len, n, seeds, 2500, 1, lanes=lanes)
        for c in range(len(seeds)):
            assert {k: many[c][k] for k in keys} == {k: one[c][k] for k in keys}, (lanes, c)
            assert np.array_equal(many_final[c][0], one_final[c][0]) and np.array_equal(many_final[c][1], one_final[c][1])
            assert many_final[c][2:] == one_final[c][2:]
        assert [b for _, b in log] == sorted((b for _, b in log), reverse=True)
        assert log[-1][1] == min(r["best_length"] for r in many)


@pytest.mark.parametrize("algorithm", [0, 2])
def test_runs_of_acceptances_in_one_step_leave_the_trajectory_alone(mods, algorithm):
    """One chain with cumulative candidates (lvbhost_anneal_params::run_levels: up to 3 or 6 accepted moves per scoring
    walk while the chain is hot, host-drawn, scored by lvbgpu_score_batch as rewrites of the resident tree, committed in
    one commit walk) follows the trajectory it follows with one move per step - same counts, lengths, trees, treestack -
    in fewer steps; the resident tree, its length and the host's mirror agree at the end (_run_chains), and so does a
    full evaluation of the final topology on a fresh context.  CPU tier: tests/test_anneal_chains_cpu.py."""
    api, host = mods
    n, m = 60, 4000
    rows, min_len = host.prepare_alignment(synth.treelike_rows(n, m, 33))
    keys = ("start_length", "best_length", "final_length", "consumed", "accepted", "temperatures", "reroots", "topologies", "t_final")
    one, one_final, _ = _run_chains(api, host, rows, min_len, n, [21], 6000, algorithm, run_levels=1)
    for levels in (3, 6):
        runs, runs_final, _ = _run_chains(api, host, rows, min_len, n, [21], 6000, algorithm, run_levels=levels)
        assert {k: runs[0][k] for k in keys} == {k: one[0][k] for k in keys}, levels
        assert all(np.array_equal(a, b) for a, b in zip(runs_final[0][:2], one_final[0][:2])) and runs_final[0][2:] == one_final[0][2:]
        assert runs[0]["device_steps"] < one[0]["device_steps"]
        # both kinds of step happened: host-drawn ones while the chain was hot, device-drawn ones before (the starting
        # temperature) and after (cooled off), so the generator's tables were made again from the host's tree on the way
        assert 0 < runs[0]["host_steps"] < runs[0]["device_steps"]
    assert one[0]["consumed"] == 6000 and one[0]["accepted"] > 100 and one[0]["reroots"] >= 5
    fresh = api.FitchContext(text_rows=rows)
    l, r, root, _ = one_final[0]
    assert fresh.set_tree(l, r, root) == one[0]["final_length"]
    fresh.close()


def test_host_made_candidates_of_several_chains_in_one_walk(mods):
    """lvbgpu_chains_score_edits / lvbgpu_chains_commit_edits against lvbgpu_select_chain + lvbgpu_score_batch /
    lvbgpu_commit chain by chain: single moves and cumulative rewrites of runs of moves, chains in any mix, the
    generator's tables following on the device (the next device draw equals the one a fresh upload gives)."""
    api, host = mods
    n, m, R = 40, 26000, 5                                         # 13 tiles: most rounds are too large for a direct step
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 17))
    both = [api.FitchContext(text_rows=rows) for _ in range(2)]
    trees = [host.HostTree(n, seed=300 + c) for c in range(R)]
    for ctx in both:
        ctx.set_chains(R)
        for c, t in enumerate(trees):
            ctx.select_chain(c)
            t.upload(ctx)
    one_by_one, together = both
    rng = np.random.default_rng(3)
    reused = 0
    for round_ in range(6):
        chains, cands = [], []
        for c in rng.permutation(R)[: int(rng.integers(1, R + 1))]:
            c = int(c)
            for _ in range(int(rng.integers(1, 40))):
                # a run of 1 .. 3 moves, given as its cumulative rewrites of the chain's tree
                copy = host.HostTree(left=trees[c].arrays()[1], right=trees[c].arrays()[2], root=trees[c].root, seed=int(rng.integers(1, 1 << 30)))
                for _ in range(int(rng.integers(1, 4))):
                    copy.apply(copy.propose(int(rng.integers(0, 3))))
                _, l0, r0 = trees[c].arrays()
                _, l1, r1 = copy.arrays()
                e = api.edits_between(l0, r0, l1, r1)
                copy.close()
                if len(e) == 0:                                       # (the run undid itself)
                    continue
                cands.append(e)
                chains.append(c)
        order = np.argsort(np.array(chains), kind="stable")          # one chain's candidates adjacent
        chains = [chains[i] for i in order]
        cands = [cands[i] for i in order]
        got = together.chains_score_edits(chains, cands)
        want = np.zeros(len(cands), dtype=np.int64)
        for c in set(chains):
            idx = [i for i, x in enumerate(chains) if x == c]
            one_by_one.select_chain(c)
            want[idx] = one_by_one.score_batch([cands[i] for i in idx])
        assert np.array_equal(got, want), round_
        # accept the best candidate of every chain that has one
        pick = {}
        for i, c in enumerate(chains):
            if c not in pick or got[i] < got[pick[c]]:
                pick[c] = i
        cs = sorted(pick)
        before = together.commits_reusing_programs()
        together.chains_commit_edits(cs, [cands[pick[c]] for c in cs])
        # (scored a moment ago, trees unchanged: the commit walks the scored programs - unless the batch was small enough
        # to be read in place from pinned memory, which the next build overwrites)
        reused += together.commits_reusing_programs() - before
        for c in cs:
            one_by_one.select_chain(c)
            assert one_by_one.commit(cands[pick[c]]) == got[pick[c]]
            trees[c].apply(cands[pick[c]])
            together.select_chain(c)
            assert together.current_length() == got[pick[c]]
            _, l, r, root = together.topology()
            _, hl, hr = trees[c].arrays()
            assert np.array_equal(l, hl) and np.array_equal(r, hr) and root == trees[c].root
        # the device's own draws from the committed trees agree on both contexts (tables rebuilt on the device / by the host)
        draws = [(c, 33, -1, 70 + round_) for c in range(R)]
        a, b = together.chains_propose_score(draws), one_by_one.chains_propose_score(draws)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert reused >= 3
    # a commit of rewrites that were NOT scored builds its programs (same result)
    e = trees[0].propose(1)
    together.chains_commit_edits([0], [e])
    one_by_one.select_chain(0)
    want = one_by_one.commit(e)
    together.select_chain(0)
    assert together.current_length() == want
    with pytest.raises(api.LvbGpuError):
        together.chains_commit_edits([0, 0], [cands[0], cands[0]])                    # a chain listed twice
    with pytest.raises(api.LvbGpuError):
        together.chains_score_edits([R], [cands[0]])                                 # no such chain
    for ctx in both:
        ctx.close()


def test_rerooting_several_chains_at_once_equals_rerooting_them_one_by_one(mods):
    """lvbgpu_chains_reroot: one commit walk for all, tables rebuilt on the device; same state as a commit of the
    re-root rewrites chain by chain, and the neighbourhoods drawn afterwards are the same."""
    api, host = mods
    n, m, R = 30, 900, 5
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 15))
    multi, ref = api.FitchContext(text_rows=rows), api.FitchContext(text_rows=rows)
    trees = [host.HostTree(n, seed=200 + c) for c in range(R)]
    for ctx in (multi, ref):
        ctx.set_chains(R)
        for c in range(R):
            ctx.select_chain(c)
            trees[c].upload(ctx)
    rng = np.random.default_rng(4)
    for step in range(8):
        # draw first, so that the device holds current tables for every chain (the rebuild-in-place case)
        got = multi.chains_propose_score([(c, 9, -1, 50 * step + c) for c in range(R)])
        who = sorted(rng.choice(R, size=int(rng.integers(1, R + 1)), replace=False).tolist())
        reqs = []
        for c in who:
            nr = int((trees[c].root + 1 + rng.integers(0, n - 1)) % n)
            reqs.append((c, nr))
            ed = trees[c].reroot_edits(nr)
            ref.select_chain(c)
            length = ref.current_length()
            assert ref.commit(ed, root=nr) == length          # a re-root keeps the length
            trees[c].apply(ed, nr)
        multi.chains_reroot(reqs)
        for c in range(R):
            multi.select_chain(c)
            ref.select_chain(c)
            assert multi.current_length() == ref.current_length()
            pm, pr = multi.topology(), ref.topology()
            assert all(np.array_equal(a, b) for a, b in zip(pm[:3], pr[:3])) and pm[3] == pr[3] == trees[c].root
            assert np.array_equal(multi.changes(), ref.changes())
            assert np.array_equal(multi.all_sets(), ref.all_sets())
            seed = 900 + 10 * step + c
            assert np.array_equal(multi.propose_score(25, -1, seed), ref.propose_score(25, -1, seed))
    # many re-roots in a row with no batch collected in between: the pinned slots their programs lie in are recycled
    # without events (a later collected batch proves the readers done), so a run of them has to drain the streams itself
    for burst in range(7):
        reqs = []
        for c in range(R):
            nr = int((trees[c].root + 1 + rng.integers(0, n - 1)) % n)
            reqs.append((c, nr))
            ed = trees[c].reroot_edits(nr)
            ref.select_chain(c)
            ref.commit(ed, root=nr)
            trees[c].apply(ed, nr)
        multi.chains_reroot(reqs)
    for c in range(R):
        multi.select_chain(c)
        ref.select_chain(c)
        assert multi.current_length() == ref.current_length()
        assert np.array_equal(multi.changes(), ref.changes()) and np.array_equal(multi.all_sets(), ref.all_sets())
        assert np.array_equal(multi.propose_score(25, -1, 4242 + c), ref.propose_score(25, -1, 4242 + c))
    with pytest.raises(api.LvbGpuError):
        multi.chains_reroot([(0, trees[0].root)])              # already the root
    with pytest.raises(api.LvbGpuError):
        multi.chains_reroot([(0, n + 2)])                      # not a leaf
    multi.close()
    ref.close()


def test_two_batches_in_flight_give_what_one_at_a_time_gives(mods):
    """lvbgpu_chains_submit / _collect: slots 0 and 1 in flight together (the second batch's generator runs beside the
    first one's walk); lengths equal lvbgpu_propose_score with the same seeds; a commit from one batch makes the other
    batch's candidates of that chain stale."""
    api, host = mods
    n, m, B = 120, 9000, 700
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 33))
    ctx = api.FitchContext(text_rows=rows)
    tree = host.HostTree(n, seed=34)
    tree.upload(ctx)
    want = [ctx.propose_score(B, -1, 500 + i) for i in range(9)]
    counts = ctx.chains_submit(0, [(0, B, -1, 500)])
    ctx.chains_submit(1, [(0, B, -1, 501)])
    with pytest.raises(api.LvbGpuError) as ei:
        ctx.chains_submit(1, [(0, B, -1, 777)])        # that slot is still in flight
    assert ei.value.status == -5
    for i in range(9):
        got = ctx.chains_collect(i % 2, counts)[0]
        assert np.array_equal(got, want[i]), i
        if i + 2 < 9:
            ctx.chains_submit(i % 2, [(0, B, -1, 500 + i + 2)])
    with pytest.raises(api.LvbGpuError):
        ctx.chains_collect(0, counts)                   # nothing in flight there any more
    # picks come from the batch collected last; the other batch goes stale with the commit
    ctx.chains_submit(0, [(0, 50, 1, 1)])
    ctx.chains_submit(1, [(0, 50, 1, 2)])
    a = ctx.chains_collect(0, [50])[0]
    b = ctx.chains_collect(1, [50])[0]
    ctx.chains_commit([(0, int(np.argmin(b)))])         # from slot 1 (collected last)
    assert ctx.current_length() == b.min()
    ctx.chains_submit(0, [(0, 50, 1, 3)])
    c = ctx.chains_collect(0, [50])[0]
    ctx.chains_commit([(0, int(np.argmin(c)))])
    assert ctx.current_length() == c.min()
    ctx.close()


def test_site_axis_shards_add_up(mods):
    """SURVEY.md 8(e), second sharding: contexts over column slices of the alignment score the same candidates and the
    slices' lengths add up to the whole alignment's (here: three slices on one GPU, then the RCCL sum with one rank)."""
    api, host = mods
    n, m = 50, 9000
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 61))
    m = len(rows[0])
    whole = api.FitchContext(text_rows=rows)
    parts = []
    for r in range(3):
        lo, hi = api.site_slice(m, r, 3)
        assert lo % 2048 == 0 and (hi % 2048 == 0 or hi == m) and lo < hi
        parts.append(api.FitchContext(text_rows=[row[lo:hi] for row in rows]))
    assert api.site_slice(m, 0, 3)[0] == 0 and api.site_slice(m, 2, 3)[1] == m
    tree = host.HostTree(n, seed=62)
    total = tree.upload(whole)
    assert sum(tree.upload(p) for p in parts) == total
    for step in range(6):
        cands = [tree.propose(k % 3) for k in range(64)]
        want = whole.score_batch(cands)
        got = sum(p.score_batch(cands) for p in parts)
        assert np.array_equal(got, want)
        pick = cands[int(np.argmin(want))]
        assert sum(p.commit(pick) for p in parts) == whole.commit(pick)
        tree.apply(pick)
    uid = api.comm_unique_id()
    parts[0].comm_init(1, 0, uid)
    v = parts[0].score_batch([tree.propose(1) for _ in range(40)])
    assert np.array_equal(parts[0].allreduce_sum(v), v)       # one rank: the sum is the value
    for p in parts:
        p.close()
    whole.close()


@pytest.mark.parametrize("pair_min", [0, 4])
def test_random_operations_over_chains_against_the_cpu_oracle(mods, monkeypatch, pair_min):
    """A random mix of everything that can change a chain - picks from multi-chain batches, batched re-roots, commits of
    host-built rewrites, device moves through the single-tree calls, host-made (cumulative) candidates of several chains
    scored in one walk and accepted in one commit walk (the accepted ones' SCORED programs walked again, or programs
    built for the commit), whole annealing steps decided and committed by the library, big batches through the walk that
    takes two candidates per wave - with every chain checked against the CPU ORACLE after each operation: full evaluation
    of the topology the library reports == the resident length, per-node changes and node sets; and the next device-drawn
    neighbourhood scores what the oracle scores.  pair_min 4: the same with LVBGPU_PAIR=4 (every device- or host-built
    batch of four candidates and more is walked two candidates per wave)."""
    from oracle import binding as ob
    from tests import helpers
    api, host = mods
    if pair_min:
        monkeypatch.setenv("LVBGPU_PAIR", str(pair_min))
    n, m, R = 26, 520, 4
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 101))
    enc = ob.encode_rows(rows)
    ctx = api.FitchContext(text_rows=rows)
    ctx.set_chains(R)
    for c in range(R):
        ctx.select_chain(c)
        host.HostTree(n, seed=300 + c).upload(ctx)

    def oracle_of(c):
        ctx.select_chain(c)
        _, l, r, root = ctx.topology()
        l64, r64 = l.astype(np.int64), r.astype(np.int64)
        t = ob.OracleTree(n, enc.shape[1], enc)
        t.set_topology(helpers.parents_of(l64, r64), l64, r64, root)
        return t, l, r, root

    def check(c):
        t, l, r, root = oracle_of(c)
        assert ctx.current_length() == t.getplen(), c
        assert np.array_equal(ctx.changes()[n:], t.changes()[n:])
        assert np.array_equal(ctx.all_sets(), t.all_sets())
        return t, l, r, root

    def full_length(l, r, root, edits):
        """length of the tree the rewrites give, from the leaves up (the oracle's getplen with every node dirty)"""
        nl, nr = helpers.apply_edits(l, r, edits)
        t = ob.OracleTree(n, enc.shape[1], enc)
        t.set_topology(helpers.parents_of(nl, nr), nl, nr, root)
        return t.getplen()

    def merged(first, second):
        """cumulative rewrites: `second` was drawn on the tree `first` gives (a later rewrite of a node replaces the earlier)"""
        out = {int(e["node"]): (int(e["left"]), int(e["right"])) for e in first}
        out.update({int(e["node"]): (int(e["left"]), int(e["right"])) for e in second})
        return np.array([(v, a, b) for v, (a, b) in out.items()], dtype=api.EDIT_DTYPE)

    def host_made(c, k, seed):
        """k candidates of chain c as rewrites of its resident tree: single moves and runs of two or three moves"""
        _, l, r, root = oracle_of(c)
        cands = []
        for i in range(k):
            ht = host.HostTree(left=l, right=r, root=root, seed=seed + 13 * i)
            e = ht.propose(int(rng.integers(0, 3)))
            cum = e
            for _ in range(int(rng.integers(0, 3))):      # a run: the next move is drawn on the tree so far
                ht.apply(e)
                e = ht.propose(int(rng.integers(0, 3)))
                cum = merged(cum, e)
            ht.close()
            cands.append(cum)
        return cands, (l, r, root)

    rng = np.random.default_rng(12)
    reused0, steps_taken, big_scored = ctx.commits_reusing_programs(), 0, 0
    for step in range(48):
        op = int(rng.integers(0, 7))
        if op == 0:      # one step over a random subset of chains, some of them accept
            active = sorted(rng.choice(R, size=int(rng.integers(1, R + 1)), replace=False).tolist())
            draws = [(c, int(rng.integers(2, 40)), -1, 10_000 + 97 * step + c) for c in active]
            lens = ctx.chains_propose_score(draws)
            picks = [(c, int(rng.integers(0, cnt))) for (c, cnt, _, _), _ in zip(draws, lens) if rng.random() < 0.7]
            want = {c: int(l[b]) for (c, b), l in ((p, lens[[d[0] for d in draws].index(p[0])]) for p in picks)}
            if picks:
                ctx.chains_commit(picks)
            for c, b in picks:
                ctx.select_chain(c)
                assert ctx.current_length() == want[c]
        elif op == 1:    # batched re-roots
            who = sorted(rng.choice(R, size=int(rng.integers(1, R + 1)), replace=False).tolist())
            reqs = []
            for c in who:
                ctx.select_chain(c)
                root = ctx.topology()[3]
                reqs.append((c, int((root + 1 + rng.integers(0, n - 1)) % n)))
            ctx.chains_reroot(reqs)
        elif op == 2:    # a host-built move committed through the single-tree call
            c = int(rng.integers(0, R))
            _, l, r, root = oracle_of(c)
            ht = host.HostTree(left=l, right=r, root=root, seed=500 + step)
            ctx.select_chain(c)
            ctx.commit(ht.propose(int(rng.integers(0, 3))))
        elif op == 3:    # a device move through the single-tree calls
            c = int(rng.integers(0, R))
            ctx.select_chain(c)
            lens = ctx.propose_score(12, -1, 777 + step)
            b = int(rng.integers(0, 12))
            edits, _ = ctx.proposal_edits(b)
            assert ctx.commit(edits) == lens[b]
        elif op == 4:    # host-made candidates of several chains (cumulative rewrites among them) in ONE walk, the accepted
            #              ones in ONE commit walk.  A big batch leaves its programs on the device and the commit walks the
            #              SCORED programs again; a small one is read in place from pinned memory and the commit builds anew
            big = bool(rng.integers(0, 2))
            who = sorted(rng.choice(R, size=int(rng.integers(1, R + 1)), replace=False).tolist())
            chain_of, cands, trees_of = [], [], {}
            for c in who:
                mine, trees_of[c] = host_made(c, int(rng.integers(150, 200)) if big else int(rng.integers(2, 7)), 900 * step + c)
                chain_of += [c] * len(mine)
                cands += mine
            lens = ctx.chains_score_edits(chain_of, cands)
            sample = range(len(cands)) if not big else rng.choice(len(cands), size=24, replace=False).tolist()
            for b in sample:
                assert int(lens[b]) == full_length(*trees_of[chain_of[b]], cands[b]), (step, b)
            take = {c: int(rng.choice([b for b in range(len(cands)) if chain_of[b] == c])) for c in who if rng.random() < 0.8}
            if take:
                before = ctx.commits_reusing_programs()
                ctx.chains_commit_edits(list(take), [cands[b] for b in take.values()])
                # (a batch beyond the direct-step size leaves its programs on the device: 4 chains x 150+ candidates; a
                # small one is a direct step read in place - unless it is walked two candidates per wave, which is not)
                reused = ctx.commits_reusing_programs() == before + 1
                assert reused if len(cands) > 512 else (pair_min > 0 or not reused), (step, len(cands), reused)
                for c, b in take.items():
                    ctx.select_chain(c)
                    assert ctx.current_length() == int(lens[b]), (step, c)
        elif op == 5:    # a whole annealing step: the library decides by decide.h's rule and commits every chain's pick
            active = sorted(rng.choice(R, size=int(rng.integers(1, R + 1)), replace=False).tolist())
            draws = [(c, int(rng.integers(1, 60)), [0, 1, 2, -1][(step + c) % 4], 5000 * step + c) for c in active]
            curs = []
            for c in active:
                ctx.select_chain(c)
                curs.append(ctx.current_length())
            rules = [(curs[i], [1e-9, 1e-5, 3e-4, 5e-2][(step + 2 * c) % 4], 3.0 * m, 17 * step + c) for i, c in enumerate(active)]
            before = {c: oracle_of(c)[1:] for c in active}
            lens, picks = ctx.chains_step(draws, rules)
            for i, c in enumerate(active):
                if picks[i] >= 0:
                    steps_taken += 1
                    edits = ctx.chains_step_edits(i)
                    assert int(lens[i][picks[i]]) == full_length(*before[c], edits), (step, c)
                    ctx.select_chain(c)
                    assert ctx.current_length() == int(lens[i][picks[i]])
                    # a worse candidate before the pick was refused, one no worse than the tree would have been taken
                    assert all(int(v) > curs[i] for v in lens[i][: picks[i]])
                else:
                    assert all(int(v) > curs[i] for v in lens[i])
        else:            # a big device-drawn batch of one chain (with LVBGPU_PAIR: two candidates per wave), sampled
            c = int(rng.integers(0, R))
            _, l, r, root = oracle_of(c)
            ctx.select_chain(c)
            B = int(rng.integers(600, 700))
            lens = ctx.propose_score(B, -1, 4242 + step)
            for b in rng.choice(B, size=24, replace=False).tolist():
                edits, _ = ctx.proposal_edits(int(b))
                assert int(lens[b]) == full_length(l, r, root, edits), (step, c, b)
            big_scored += 1
        for c in range(R):
            cur, l, r, root = check(c)
            # the next neighbourhood of this chain, drawn on the device, against the oracle's incremental getplen
            lens = ctx.propose_score(6, -1, 31 * step + c)
            for b in range(6):
                edits, _ = ctx.proposal_edits(b)
                ht = host.HostTree(left=l, right=r, root=root)
                prog = ht.program(mode=0, edits=edits)
                nl, nr = helpers.apply_edits(l, r, edits)
                cand = ob.OracleTree(n, enc.shape[1])
                cand.copy_from(cur)
                cand.set_topology(helpers.parents_of(nl, nr), nl, nr, root)
                cand.mark_dirty([d for d in prog["dsts"] if d >= 0])
                assert int(lens[b]) == cand.getplen(), (step, c, b)
    # every kind of operation happened, both branches of the host-made commit among them
    assert ctx.commits_reusing_programs() > reused0 and steps_taken > 0 and big_scored > 0
    assert (ctx.paired_walks() > 0) == (pair_min > 0)
    ctx.close()


@pytest.mark.parametrize("n,m", [(48, 4000), (1500, 260)])
def test_commits_reroots_and_the_next_generator_in_one_launch(mods, n, m):
    """What lies between two scoring walks of an annealing step - the commit walk of the chains' accepted candidates (their
    own device-built programs), the re-roots of other chains (host-built programs), the table rebuilds of both and the NEXT
    step's generator, whose segments wait for their chains' rebuilds - goes out as ONE post launch when the steps take
    turns in the two batch slots.  Same lengths, picks, trees, per-node changes and node sets as a context that is made to
    catch up after every single call (its commits, re-roots and generators are launches of their own), and as the CPU
    oracle's full evaluation of the final trees.  1500 taxa: trees whose new tables do not fit LDS beside the
    rebuild's own arrays (every entry stored where it belongs, no draw by the rebuilding workgroup)."""
    from oracle import binding as ob
    from tests import helpers
    api, host = mods
    R = 6 if n < 100 else 3
    rows, min_len = host.prepare_alignment(synth.treelike_rows(n, m, 77))
    enc = ob.encode_rows(rows)
    one, ref = api.FitchContext(text_rows=rows), api.FitchContext(text_rows=rows)
    cur = []
    for c_ in (one, ref):
        c_.set_chains(R)
    for c in range(R):
        t = host.HostTree(n, seed=4100 + c)
        for c_ in (one, ref):
            c_.select_chain(c)
            length = t.upload(c_)
        cur.append(length)
    roots = []
    for c in range(R):
        one.select_chain(c)
        roots.append(one.topology()[3])
    rng = np.random.default_rng(8)
    both = rerooted = 0
    nsteps = 60 if n < 100 else 24
    for step in range(nsteps):
        slot = step & 1
        active = sorted(rng.choice(R, size=int(rng.integers(2, R + 1)), replace=False).tolist())
        # (now and then a draw too large for the rebuilding workgroup to make itself: its generator workgroups wait)
        draws = [(c, int(rng.integers(1, 70)) if (step + c) % 5 else int(rng.integers(100, 300)), [1, 2, -1, 0][(step + c) % 4],
                  9000 * step + c) for c in active]
        rules = [(cur[c], [1e-9, 2e-5, 4e-4, 5e-2][(step + c) % 4], float(min_len), 31 * step + c) for c in active]
        lens, picks = one.chains_step(draws, rules, slot=slot)
        ref.synchronize()
        rlens, rpicks = ref.chains_step(draws, rules, slot=slot)
        ref.synchronize()
        assert np.array_equal(picks, rpicks), step
        for i, c in enumerate(active):
            assert np.array_equal(lens[i], rlens[i]), (step, c)
            if picks[i] >= 0:
                cur[c] = int(lens[i][picks[i]])
        # re-roots of chains that accepted nothing (the usual case in a run: a chain's tick comes when a whole draw was
        # refused) and now and then of one that did (its accepted move has to reach the host first: two launches)
        who = [c for i, c in enumerate(active) if (picks[i] < 0 or rng.random() < 0.15) and rng.random() < 0.6]
        if who:
            reqs = []
            for c in who:
                roots[c] = int((roots[c] + 1 + rng.integers(0, n - 1)) % n)
                reqs.append((c, roots[c]))
            both += sum(1 for i, c in enumerate(active) if c in who and picks[i] >= 0)
            rerooted += len(who)
            one.chains_reroot(reqs)
            ref.chains_reroot(reqs)
            ref.synchronize()
    posts, with_generator = one.post_launches()
    rposts, rwith = ref.post_launches()
    assert with_generator > nsteps // 2 and rwith == 0, (posts, with_generator, rposts, rwith)
    assert posts < rposts and rerooted > nsteps // 6 and (both > 0 or n > 100)
    for c in range(R):
        for c_ in (one, ref):
            c_.select_chain(c)
        assert one.current_length() == ref.current_length() == cur[c], c
        assert np.array_equal(one.changes(), ref.changes())
        assert np.array_equal(one.all_sets(), ref.all_sets())
        pm, pr = one.topology(), ref.topology()
        assert all(np.array_equal(a, b) for a, b in zip(pm[:3], pr[:3])) and pm[3] == pr[3] == roots[c]
        _, l, r, root = pm
        l64, r64 = l.astype(np.int64), r.astype(np.int64)
        t = ob.OracleTree(n, enc.shape[1], enc)
        t.set_topology(helpers.parents_of(l64, r64), l64, r64, root)
        assert t.getplen() == cur[c]
        assert np.array_equal(one.changes()[n:], t.changes()[n:]) and np.array_equal(one.all_sets(), t.all_sets())
        # ... and the generator's tables have followed: the next neighbourhoods are the same
        assert np.array_equal(one.propose_score(40, -1, 99 + c), ref.propose_score(40, -1, 99 + c))
    one.close()
    ref.close()


def test_pick_slots_survive_collects_of_batches_submitted_before_their_use(mods):
    """a collected batch proves a pinned pick slot's readers done only if it was SUBMITTED AFTER the slot was
    used.  Here a big batch is submitted first, then re-roots and commits pile up behind it (their commit walks read
    programs in place from the pinned slots), the old batch is collected in between (which used to reset the recycling
    count), and more re-roots follow - enough to come round to the first slot again while its walk may still be queued.
    Every chain must end exactly where the same re-roots, done one by one with a synchronous commit, put it."""
    api, host = mods
    n, m, R = 64, 20000, 4
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 23))
    multi, ref = api.FitchContext(text_rows=rows), api.FitchContext(text_rows=rows)
    trees = [host.HostTree(n, seed=700 + c) for c in range(R)]
    for ctx in (multi, ref):
        ctx.set_chains(R)
        for c in range(R):
            ctx.select_chain(c)
            trees[c].upload(ctx)
    rng = np.random.default_rng(8)

    def reroot_all():
        reqs = []
        for c in range(1, R):                      # chain 0 is the one the long batch was drawn from: left alone
            nr = int((trees[c].root + 1 + rng.integers(0, n - 1)) % n)
            reqs.append((c, nr))
            ed = trees[c].reroot_edits(nr)
            ref.select_chain(c)
            ref.commit(ed, root=nr)
            trees[c].apply(ed, nr)
        multi.chains_reroot(reqs)

    for rnd in range(3):
        counts = multi.chains_submit(0, [(0, 6000, 2, 40 + rnd)])   # long: its walk keeps the stream busy for a while
        reroot_all()                                                # slot p0, queued behind the batch
        reroot_all()                                                # p1
        got = multi.chains_collect(0, counts)[0]                    # submitted BEFORE those uses: proves nothing about them
        ref.select_chain(0)
        assert np.array_equal(got, ref.propose_score(6000, 2, 40 + rnd))
        for _ in range(4):                                          # p2, p3, then p0 and p1 again
            reroot_all()
    for c in range(R):
        multi.select_chain(c)
        ref.select_chain(c)
        assert multi.current_length() == ref.current_length()
        assert np.array_equal(multi.changes(), ref.changes()) and np.array_equal(multi.all_sets(), ref.all_sets())
        assert np.array_equal(multi.propose_score(30, -1, 99 + c), ref.propose_score(30, -1, 99 + c))
    multi.close()
    ref.close()


def test_commit_while_the_other_slot_holds_the_same_chain(mods):
    """include/lvbgpu.h lets a chain sit in both slots.  A commit picked from the batch collected first
    rebuilds that chain's generator tables on the side stream - which must not overtake the OTHER batch's generator,
    still queued on the main stream and reading the same tables.  The other batch was drawn from the tree BEFORE the
    commit: its lengths must be exactly what that tree gives (and picking from it afterwards is refused as stale)."""
    api, host = mods
    n, m = 500, 50000
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 3))
    ctx, ref = api.FitchContext(text_rows=rows), api.FitchContext(text_rows=rows)
    tree = host.HostTree(n, seed=77)
    tree.upload(ctx)
    tree.upload(ref)
    for rnd in range(12):
        B0, B1 = 4096, 2048
        want1 = ref.propose_score(B1, 1, 9000 + rnd)               # the old tree's neighbourhood
        want0 = ref.propose_score(B0, 1, 8000 + rnd)
        c0 = ctx.chains_submit(0, [(0, B0, 1, 8000 + rnd)])
        got0 = ctx.chains_collect(0, c0)[0]
        # two more batches queue up: slot 0 keeps the stream busy, so slot 1's generator is still waiting when the commit comes
        ctx.chains_submit(0, [(0, B0, 1, 8100 + rnd)])
        c1 = ctx.chains_submit(1, [(0, B1, 1, 9000 + rnd)])
        assert np.array_equal(got0, want0)
        with pytest.raises(api.LvbGpuError):
            ctx.chains_commit([(0, 0)])                             # slot 0's NEW batch is in flight: nothing to pick from
        junk = ctx.chains_collect(0, c0)[0]
        b = int(np.argmin(junk))
        ctx.chains_commit([(0, b)])                                 # rebuild on the side stream; slot 1 still in flight
        got1 = ctx.chains_collect(1, c1)[0]
        assert np.array_equal(got1, want1), rnd                     # drawn from intact tables of the OLD tree
        with pytest.raises(api.LvbGpuError) as ei:
            ctx.chains_commit([(0, 0)])                             # ... and stale now
        assert ei.value.status == -5
        # the reference context follows the same move
        ref.propose_score(B0, 1, 8100 + rnd)
        edits, _ = ref.proposal_edits(b)
        assert ref.commit(edits) == junk[b] == ctx.current_length()
    assert np.array_equal(ctx.all_sets(), ref.all_sets())
    ctx.close()
    ref.close()


def test_proposal_edits_are_refused_once_a_chain_commit_has_moved_the_tree(mods):
    """after lvbgpu_propose_score -> lvbgpu_chains_commit (tables rebuilt on the device, versions equal
    again) the OLD batch's rewrites are relative to a tree that is gone."""
    api, host = mods
    n, m = 40, 1200
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 5))
    ctx = api.FitchContext(text_rows=rows)
    tree = host.HostTree(n, seed=6)
    tree.upload(ctx)
    lens = ctx.propose_score(16, -1, 3)
    edits, _ = ctx.proposal_edits(2)                                # fine: same tree
    assert len(edits) >= 2
    ctx.chains_commit([(0, int(np.argmin(lens)))])
    with pytest.raises(api.LvbGpuError) as ei:
        ctx.proposal_edits(2)
    assert ei.value.status == -5
    lens = ctx.propose_score(16, -1, 4)
    ctx.proposal_edits(1)
    nr = (ctx.topology()[3] + 3) % n
    ctx.chains_reroot([(0, nr)])
    with pytest.raises(api.LvbGpuError) as ei:
        ctx.proposal_edits(1)
    assert ei.value.status == -5
    ctx.close()


def test_cfg2_chains_step_equals_single_chain_steps(mods):
    """BASELINE configs[1] (64 x 10 000, NNI, 1024 candidates) through the multi-chain step - R chains x 1024 candidates in
    one generator launch and one walk, which is how bench.py fills the chip at this shape - gives every chain exactly the
    lengths its own lvbgpu_propose_score step gives (same seed => same moves), two steps in flight included."""
    api, host = mods
    n, m, R, B = 64, 10000, 4, 1024
    rows, _ = host.prepare_alignment(synth.treelike_rows(n, m, 3))
    multi, ref = api.FitchContext(text_rows=rows), api.FitchContext(text_rows=rows)
    multi.set_chains(R)
    ref.set_chains(R)
    for c in range(R):
        t = host.HostTree(n, seed=300 + c)
        for ctx in (multi, ref):
            ctx.select_chain(c)
            t.upload(ctx)
    counts = multi.chains_submit(0, [(c, B, 0, 10 + c) for c in range(R)])
    multi.chains_submit(1, [(c, B, 0, 50 + c) for c in range(R)])
    for slot, base in ((0, 10), (1, 50)):
        got = multi.chains_collect(slot, counts)
        for c in range(R):
            ref.select_chain(c)
            assert np.array_equal(got[c], ref.propose_score(B, 0, base + c)), (slot, c)
    multi.close()
    ref.close()


def _take(length, cur, t, minlen, seed, j):
    """lvb_amd/csrc/decide.h, restated: -> (taken?, margin) - margin = |u - p| where a draw decides (borderline cases may
    differ by an ulp of exp between host and device)"""
    import math
    M = (1 << 64) - 1
    if length <= 0 or length >= (1 << 61):
        return False, 1.0
    if length <= cur:
        return True, 1.0
    deltah = min(1.0, minlen / cur - minlen / length)
    if -deltah < t * -25.328436022934504:
        return False, 1.0
    z = (seed + 0x9E3779B97F4A7C15 * (j + 1)) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    z ^= z >> 31
    u = (z >> 11) * (1.0 / 9007199254740992.0)
    p = math.exp(-deltah / t)
    return u < p, abs(u - p)


def test_a_step_decides_and_commits_like_the_host_would(mods):
    """lvbgpu_chains_step_*: the accept decision rides with the batch.  Lengths are the plain step's; every chain's pick is
    the FIRST candidate decide.h's rule takes (checked against a restatement of the rule here); the accepted moves are
    committed - resident lengths, per-node changes, node sets, topologies and the next neighbourhoods equal a reference
    context that committed the same picks through lvbgpu_chains_commit; the moves' rewrites are the candidates' own."""
    api, host = mods
    n, m, R = 60, 5000, 5
    rows, min_len = host.prepare_alignment(synth.treelike_rows(n, m, 41))
    ctx, ref = api.FitchContext(text_rows=rows), api.FitchContext(text_rows=rows)
    cur = []
    for c_ in (ctx, ref):
        c_.set_chains(R)
    for c in range(R):
        t = host.HostTree(n, seed=900 + c)
        for c_ in (ctx, ref):
            c_.select_chain(c)
            length = t.upload(c_)
        cur.append(length)
    rng = np.random.default_rng(5)
    accepted = nothing = 0
    for step in range(40):
        active = sorted(rng.choice(R, size=int(rng.integers(1, R + 1)), replace=False).tolist())
        draws = [(c, int(rng.integers(1, 90)), [0, 1, 2, -1][(step + c) % 4], 7000 * step + c) for c in active]
        # temperatures from "nothing worse is ever taken" to "nearly everything is"
        # (every third rule claims a current length of 1: everything is worse and, that cold, nothing is taken)
        rules = [(1, 1e-12, float(min_len), 5) if (step + c) % 3 == 0 else
                 (cur[c], [1e-9, 1e-5, 3e-4, 5e-2][(step + 2 * c) % 4], float(min_len), 111 * step + c) for c in active]
        lens, picks = ctx.chains_step(draws, rules)
        want_lens = ref.chains_propose_score(draws)
        commit = []
        for i, (c, count, kind, seed) in enumerate(draws):
            assert np.array_equal(lens[i], want_lens[i]), (step, c)
            first, sure = -1, True
            for j in range(count):
                take, margin = _take(int(lens[i][j]) if lens[i][j] < np.iinfo(np.int64).max else 1 << 61, *rules[i], j)
                if margin < 1e-9:
                    sure = False                      # an ulp of exp could turn this one: accept the device's word
                    break
                if take:
                    first = j
                    break
            if sure:
                assert picks[i] == first, (step, c, picks[i], first)
            if picks[i] >= 0:
                accepted += 1
                commit.append((c, int(picks[i])))
                cur[c] = int(lens[i][picks[i]])
            else:
                nothing += 1
        if commit:
            ref.chains_commit(commit)
        for i, (c, count, kind, seed) in enumerate(draws):
            if picks[i] >= 0:
                e = ctx.chains_step_edits(i)
                assert len(e) >= 2
            else:
                with pytest.raises(api.LvbGpuError):
                    ctx.chains_step_edits(i)
        for c in range(R):
            ctx.select_chain(c)
            ref.select_chain(c)
            assert ctx.current_length() == ref.current_length() == cur[c], (step, c)
            assert np.array_equal(ctx.changes(), ref.changes())
            pm, pr = ctx.topology(), ref.topology()
            assert all(np.array_equal(a, b) for a, b in zip(pm[:3], pr[:3])) and pm[3] == pr[3]
            if step % 8 == 7:
                assert np.array_equal(ctx.all_sets(), ref.all_sets())
                assert np.array_equal(ctx.propose_score(20, -1, 5 + c), ref.propose_score(20, -1, 5 + c))
    assert accepted > 20 and nothing > 10                 # both outcomes were exercised
    # a step's slot cannot be collected as a plain batch, and a plain batch not as a step
    ctx.select_chain(0)
    d = ctx._draws([(0, 8, 1, 1)])
    r = np.zeros(1, dtype=api.RULE_DTYPE)
    r[0]["cur_length"], r[0]["temperature"], r[0]["min_len_tree"], r[0]["accept_seed"] = cur[0], 1e-9, float(min_len), 3
    ctx._chk(ctx.lib.lvbgpu_chains_step_submit(ctx.h, 0, 1, d.ctypes.data, r.ctypes.data))
    out, pk = np.zeros(8, dtype=np.int64), np.zeros(1, dtype=np.int32)
    assert ctx.lib.lvbgpu_chains_collect(ctx.h, 0, out) == -5
    ctx._chk(ctx.lib.lvbgpu_chains_step_collect(ctx.h, 0, out, pk))
    ctx.chains_submit(0, [(0, 8, 1, 2)])
    assert ctx.lib.lvbgpu_chains_step_collect(ctx.h, 0, out, pk) == -5
    ctx.chains_collect(0, [8])
    ctx.close()
    ref.close()


@pytest.mark.parametrize("pair", ["0", "4"])
def test_a_generator_that_waits_in_vain_gives_up_and_nothing_hangs(mods, pair):
    """The post launch's generating workgroups wait for a word of an earlier workgroup of the same launch (their chain's
    tables are out).  Every wait in a kernel is bounded; what happens when this one runs out is otherwise never seen:
    LVBGPU_DEBUG_WITHHOLD_READY (a child process - tests/children/withheld_tables.py) keeps the words back.  The candidates
    of the chains concerned come back as "not proposals", the other chains of the launch are served, nothing hangs, the
    resident trees stay what the oracle says - with one and (pair = 4) with two candidates per wave, where the workgroup
    that gives up still has to say who walks with whom."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LVBGPU_DEBUG_WITHHOLD_READY="1", LVBGPU_PAIR=pair)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "children", "withheld_tables.py")], cwd=root, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "withheld ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
