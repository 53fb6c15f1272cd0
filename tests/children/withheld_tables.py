"""Child of tests/test_gpu_chains.py::test_a_generator_that_waits_in_vain_gives_up_and_nothing_hangs (the environment is
read when the library first needs it, so the switch gets a process of its own).  LVBGPU_DEBUG_WITHHOLD_READY: the post
launch's rebuilding workgroups never say that their chain's tables are out, and the generating workgroups that wait for
them look 4096 times and give up.  Expected: their candidates come back as "not proposals" (INT64_MAX, never accepted), every
other chain of the same launch is served, nothing hangs, and the resident trees are what the CPU oracle says they are."""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from lvb_amd import api, host   # noqa: E402
from oracle import binding as ob   # noqa: E402
from tests import helpers, synth   # noqa: E402

n, m, R = 40, 2500, 3
rows, min_len = host.prepare_alignment(synth.treelike_rows(n, m, 61))
enc = ob.encode_rows(rows)
ctx = api.FitchContext(text_rows=rows)
ctx.set_chains(R)
cur = []
for c in range(R):
    ctx.select_chain(c)
    cur.append(host.HostTree(n, seed=700 + c).upload(ctx))
BIG = np.iinfo(np.int64).max
PAIRED = int(os.environ.get("LVBGPU_PAIR", "0") or 0) > 0   # (with two candidates per wave no rebuilding workgroup draws itself)
gave_up = served = 0
moved_before = set()
for step in range(10):
    # chains 0 and 1 draw more than a rebuilding workgroup draws itself (their segments go to generating workgroups, which
    # wait for the tables of a chain that moved in the step before); chain 2 draws few (its rebuilding workgroup draws them,
    # unless candidates are walked two per wave: pairs are made by generating workgroups, so its segment waits as well)
    draws = [(0, 150, 1, 10 * step), (1, 90, 2, 10 * step + 1), (2, 20, 1, 10 * step + 2)]
    # (a hot temperature: something is accepted in nearly every draw that was served)
    rules = [(cur[c], 5e-2, float(min_len), 31 * step + c) for c in range(R)]
    lens, picks = ctx.chains_step(draws, rules, slot=step & 1)
    moved_now = set()
    for c in range(R):
        if c in moved_before and (c < 2 or PAIRED):
            assert (lens[c] == BIG).all() and picks[c] < 0, (step, c)   # waited in vain: not proposals, nothing accepted
            gave_up += 1
        else:
            assert (lens[c] < BIG).all() and lens[c].min() > 0, (step, c)
            served += 1
        if picks[c] >= 0:
            cur[c] = int(lens[c][picks[c]])
            moved_now.add(c)
    moved_before = moved_now
assert gave_up >= 4 and served >= 12, (gave_up, served)
for c in range(R):
    ctx.select_chain(c)
    _, l, r, root = ctx.topology()
    l64, r64 = l.astype(np.int64), r.astype(np.int64)
    t = ob.OracleTree(n, enc.shape[1], enc)
    t.set_topology(helpers.parents_of(l64, r64), l64, r64, root)
    assert ctx.current_length() == t.getplen() == cur[c], c
    assert np.array_equal(ctx.all_sets(), t.all_sets())
    # the tables did follow (only the word was withheld): a fresh neighbourhood scores what the oracle scores
    lens = ctx.propose_score(6, 1, 99 + c)
    for b in range(6):
        edits, _ = ctx.proposal_edits(b)
        nl, nr = helpers.apply_edits(l, r, edits)
        cand = ob.OracleTree(n, enc.shape[1], enc)
        cand.set_topology(helpers.parents_of(nl, nr), nl, nr, root)
        assert int(lens[b]) == cand.getplen(), (c, b)
ctx.close()
print(f"withheld ok: {gave_up} draws given up, {served} served")
