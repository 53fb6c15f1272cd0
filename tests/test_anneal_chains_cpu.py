"""lvbhost_anneal_chains (lvb_amd/csrc/anneal_chains.cpp) as HOST LOGIC in the CPU tier: the chain state machines run
against the scorer's test double (tests/cpu_double: the oracle scores, the host library's own generators draw the
neighbours - candidate j of a draw is a function of (seed, j), as the C-ABI promises).  What must hold without a GPU:
a chain's trajectory does not depend on how many chains run beside it, the host's mirror of every chain IS the
scorer's tree, lengths only fall in the shared log, and a frozen chain stops asking.  The same properties on the HIP
scorer: tests/test_gpu_chains.py."""
import ctypes as C

import numpy as np
import pytest

from tests import synth


@pytest.fixture(scope="module")
def double():
    from oracle import binding
    binding.load_oracle()
    from tests.cpu_double import build
    lib, new_ctx, free_ctx = build.load()
    lib.lvbgpu_select_chain.argtypes = [C.c_void_p, C.c_int32]
    lib.lvbgpu_current_length.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    return lib, new_ctx, free_ctx


def run_chains(double, rows, min_len, n, seeds, max_proposals, algorithm, batch=64, run_levels=0, lanes=1):
    from lvb_amd import host
    lib, new_ctx, free_ctx = double
    ctx = new_ctx(rows)
    trees = [host.HostTree(n, seed=1000 + s, lib=lib) for s in seeds]
    params = []
    for s in seeds:
        p = host.anneal_defaults(lib)
        p.seed = 7000 + s
        p.algorithm = algorithm
        p.batch = batch
        p.t0 = 0.0                       # every chain estimates its own starting temperature
        p.min_len_tree = min_len
        p.max_proposals = max_proposals
        p.log_cap = 64
        p.run_levels = run_levels
        p.lanes = lanes
        params.append(p)
    try:
        res, log = host.anneal_chains(ctx, trees, params, lib=lib)
        final = []
        for c, t in enumerate(trees):
            if lanes == 1:                                             # (with more lanes the other lanes' contexts are gone)
                assert lib.lvbgpu_select_chain(ctx, c) == 0
                length = C.c_int64()
                assert lib.lvbgpu_current_length(ctx, C.byref(length)) == 0
                assert length.value == res[c]["final_length"]          # the resident length is what the chain believes
            _, l, r = t.arrays()
            final.append((l.copy(), r.copy(), t.root, t.best_count()))
    finally:
        for t in trees:
            t.close()
        free_ctx(ctx)
    return res, final, log


KEYS = ("start_length", "best_length", "final_length", "consumed", "accepted", "temperatures", "device_steps", "scored",
        "reroots", "topologies", "t_final")


@pytest.mark.parametrize("algorithm", [0, 1, 2])
def test_a_chains_trajectory_does_not_depend_on_the_chains_beside_it(double, algorithm):
    from lvb_amd import host
    lib = double[0]
    n, m = 24, 600
    rows, min_len = host.prepare_alignment(synth.treelike_rows(n, m, 91), lib)
    seeds = [3, 4, 5, 6, 7]
    many, many_final, log = run_chains(double, rows, min_len, n, seeds, 1500, algorithm)
    for pick in (0, 4):
        one, one_final, _ = run_chains(double, rows, min_len, n, [seeds[pick]], 1500, algorithm)
        assert {k: one[0][k] for k in KEYS} == {k: many[pick][k] for k in KEYS}
        assert all(np.array_equal(a, b) for a, b in zip(one_final[0][:2], many_final[pick][:2]))
        assert one_final[0][2:] == many_final[pick][2:]
    assert all(r["consumed"] == 1500 and r["best_length"] <= r["start_length"] for r in many)
    assert all(r["reroots"] >= 1 for r in many)                     # one per 1000 proposals and per temperature sample
    assert [b for _, b in log] == sorted((b for _, b in log), reverse=True)    # the shared log only ever improves
    assert log[-1][1] == min(r["best_length"] for r in many)


@pytest.mark.parametrize("lanes,run_levels", [(2, 0), (3, 2)])
def test_lanes_leave_every_chains_trajectory_alone(double, lanes, run_levels):
    """lvbhost_anneal_params::lanes: the chains dealt to contexts of their own (lvbgpu_fork) that one host thread serves in
    turn.  Every chain's counters, final tree and kept trees are what one lane gives it - device-drawn and with runs of
    accepted moves -, the shared log only improves and ends at the best length over all chains."""
    from lvb_amd import host
    lib = double[0]
    n, m = 16, 300
    rows, min_len = host.prepare_alignment(synth.treelike_rows(n, m, 44), lib)
    seeds = [11, 12, 13, 14, 15]
    one, one_final, _ = run_chains(double, rows, min_len, n, seeds, 500, 1, run_levels=run_levels)
    many, many_final, log = run_chains(double, rows, min_len, n, seeds, 500, 1, run_levels=run_levels, lanes=lanes)
    for c in range(len(seeds)):
        assert {k: many[c][k] for k in KEYS} == {k: one[c][k] for k in KEYS}, c
        assert np.array_equal(many_final[c][0], one_final[c][0]) and np.array_equal(many_final[c][1], one_final[c][1])
        assert many_final[c][2:] == one_final[c][2:]
    assert [b for _, b in log] == sorted((b for _, b in log), reverse=True)
    assert log[-1][1] == min(r["best_length"] for r in many)


def test_chains_run_to_the_freezing_criterion_and_stop(double):
    """No proposal cap: every chain anneals until the reference's criterion freezes it (Solve.c:409-443), the run ends
    when the last one has, and a frozen chain's counters stand still while the others go on."""
    from lvb_amd import host
    lib = double[0]
    n, m = 12, 300
    rows, min_len = host.prepare_alignment(synth.treelike_rows(n, m, 17), lib)
    res, final, log = run_chains(double, rows, min_len, n, [1, 2, 3], 0, 1, batch=32)
    assert all(r["frozen"] == 1 for r in res)
    assert len({r["consumed"] for r in res}) > 1                    # they froze at different times
    assert all(r["best_length"] <= r["start_length"] and r["final_length"] >= r["best_length"] for r in res)
    assert all(f[3] >= 1 for f in final)                            # every chain holds at least its best tree


@pytest.mark.parametrize("algorithm", [0, 1, 2])
def test_runs_of_acceptances_in_one_step_leave_the_trajectory_alone(double, algorithm):
    """while a lone chain accepts most of what it sees its candidates are cumulative (level d drawn
    on the tree the first alternatives of the levels before leave: lvbhost_anneal_params::run_levels), so one scoring
    walk advances it by a RUN of accepted moves.  Proposal i's draw and its Metropolis draw are functions of (seed, i,
    the tree it is drawn on): the trajectory with runs of up to 3 or 5 moves per step is the one with one move per step
    (run_levels = 1: the same host-drawn mode, one level) - same counts, same lengths, same trees, same treestack -
    in fewer steps.  (The device-drawn mode, run_levels = 0, draws by another law: a different, equally valid run.)"""
    from lvb_amd import host
    lib = double[0]
    n, m = 30, 900
    rows, min_len = host.prepare_alignment(synth.treelike_rows(n, m, 57), lib)
    keys = tuple(k for k in KEYS if k not in ("device_steps", "scored"))
    one, one_final, _ = run_chains(double, rows, min_len, n, [11], 2500, algorithm, run_levels=1)
    for levels in (3, 5):
        runs, runs_final, _ = run_chains(double, rows, min_len, n, [11], 2500, algorithm, run_levels=levels)
        assert {k: runs[0][k] for k in keys} == {k: one[0][k] for k in keys}, levels
        assert all(np.array_equal(a, b) for a, b in zip(runs_final[0][:2], one_final[0][:2]))
        assert runs_final[0][2:] == one_final[0][2:]
        assert runs[0]["device_steps"] < one[0]["device_steps"]                 # runs happened: fewer steps for the same moves
    assert one[0]["consumed"] == 2500 and one[0]["accepted"] > 50 and one[0]["reroots"] >= 2
    dev, _, _ = run_chains(double, rows, min_len, n, [11], 2500, algorithm, run_levels=0)
    assert dev[0]["consumed"] == 2500                                            # the device-drawn law: another valid run


def test_runs_among_other_chains_are_the_runs_of_the_chain_alone(double):
    """With run_levels > 0 the hot chains of a step share ONE scoring walk (lvbgpu_chains_score_edits) and ONE commit
    walk (lvbgpu_chains_commit_edits) while the others go through the device step: a chain's trajectory is still its
    own - the same with five chains beside it as alone, and the same for every run length."""
    from lvb_amd import host
    lib = double[0]
    n, m = 24, 600
    rows, min_len = host.prepare_alignment(synth.treelike_rows(n, m, 91), lib)
    seeds = [3, 4, 5, 6, 7]
    keys = tuple(k for k in KEYS if k not in ("device_steps", "scored"))
    many, many_final, log = run_chains(double, rows, min_len, n, seeds, 1500, 0, run_levels=3)
    for pick in (1, 4):
        for levels in (1, 3):
            one, one_final, _ = run_chains(double, rows, min_len, n, [seeds[pick]], 1500, 0, run_levels=levels)
            assert {k: one[0][k] for k in keys} == {k: many[pick][k] for k in keys}, (pick, levels)
            assert all(np.array_equal(a, b) for a, b in zip(one_final[0][:2], many_final[pick][:2]))
            assert one_final[0][2:] == many_final[pick][2:]
    assert all(r["consumed"] == 1500 for r in many)
    assert [b for _, b in log] == sorted((b for _, b in log), reverse=True)
