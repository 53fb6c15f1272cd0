"""GPU parity: the HIP path (through the C ABI) against the checkers, bit for bit.

Checker = the real reference compiled into oracle/_ref/liblvbref.so when it travelled with the
snapshot, otherwise our C restatement (oracle/fitch_oracle.c, itself pinned to the reference by
tests/test_oracle_*.py).  Integer work: the bar is exact equality of lengths, per-node
`changes` and every 64-bit state-set word.
"""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from lvb_amd import api as a
    assert a.device_count() >= 1, "no MI355X visible"
    return a


@pytest.fixture(scope="module")
def ob():
    from oracle import binding
    return binding


def _ref_or_skip(ob):
    if ob.load_ref() is None:
        pytest.skip("oracle/_ref/liblvbref.so did not travel")


def _i32(a):
    return np.asarray(a, dtype=np.int32)


# shapes: (n, m, seed).  62/984 = the stock example's post-cut shape class; 2049 sites = one word
# over a 128-word tile; 64x10k = BASELINE configs[1]
SHAPES = [(5, 2, 1), (8, 17, 2), (20, 100, 3), (33, 2049, 4), (100, 984, 5), (64, 10000, 6)]


@pytest.mark.parametrize("n,m,seed", SHAPES)
def test_full_evaluation_matches_reference(api, ob, n, m, seed):
    _ref_or_skip(ob)
    rows = synth.treelike_rows(n, m, seed) if m > 4 else synth.uniform_rows(n, m, seed)
    rr = ob.RefRun(rows=rows, seed=seed)
    try:
        ctx = api.FitchContext(rr.enc())
        for trial in range(3):
            if trial:
                rr.random_tree()
            expect = rr.getplen(0)
            p, l, r, ch, _ = rr.tree(0)
            got = ctx.set_tree(_i32(l), _i32(r), rr.root(0))
            assert got == expect
            assert ctx.current_length() == expect
            assert np.array_equal(ctx.changes()[rr.n:], ch[rr.n:])
            assert np.array_equal(ctx.all_sets(), rr.all_sets(0))
        ctx.close()
    finally:
        rr.close()


@pytest.mark.parametrize("n,m,seed", SHAPES[1:])
def test_incremental_batches_and_commits_match_reference(api, ob, n, m, seed):
    """An SA-like walk: every proposal scored in batches, some accepted (commit), re-rooted."""
    _ref_or_skip(ob)
    rows = synth.treelike_rows(n, m, seed)
    rr = ob.RefRun(rows=rows, seed=seed)
    try:
        ctx = api.FitchContext(rr.enc())
        assert rr.getplen(0) == ctx.set_tree(_i32(rr.tree(0)[1]), _i32(rr.tree(0)[2]), rr.root(0))
        B = 24
        for step in range(6):
            _, cl, cr, _, _ = rr.tree(0)
            cands, expect = [], []
            for b in range(B):
                rr.mutate(b % 3)
                _, nl, nr, _, dirty = rr.tree(1)
                cands.append(api.edits_between(cl, cr, nl, nr))
                expect.append(rr.getplen(1))
            got = ctx.score_batch(cands)
            assert np.array_equal(got, np.array(expect)), f"step {step}"
            # accept the last proposal (reference: SwapTrees), commit it on the device
            rr.swap()
            length = ctx.commit(cands[-1])
            assert length == expect[-1]
            _, _, _, ch, _ = rr.tree(0)
            assert np.array_equal(ctx.changes()[rr.n:], ch[rr.n:])
            assert np.array_equal(ctx.all_sets(), rr.all_sets(0))
            if step % 2 == 1:
                # arbreroot marks everything dirty in the reference; here it is one more edit
                _, cl, cr, _, _ = rr.tree(0)
                rr.arbreroot()
                _, nl, nr, _, _ = rr.tree(0)
                ed = api.edits_between(cl, cr, nl, nr)
                expect_len = rr.getplen(0)
                # score it as a candidate first, then commit
                assert ctx.score_batch([ed], roots=[rr.root(0)])[0] == expect_len
                assert ctx.commit(ed, root=rr.root(0)) == expect_len
                _, _, _, ch, _ = rr.tree(0)
                assert np.array_equal(ctx.changes()[rr.n:], ch[rr.n:])
                assert np.array_equal(ctx.all_sets(), rr.all_sets(0))
        ctx.close()
    finally:
        rr.close()


def test_big_batch_keeps_the_callers_order(api, ob):
    """From 2048 candidates on the library lays a resident batch out longest program first (the launch's tail)
    and un-permutes the lengths on the way back, and lvbgpu_score_batch pipelines its pieces: 3000 mixed
    NNI/SPR/TBR neighbours, each against the reference."""
    _ref_or_skip(ob)
    n, m, seed, B = 33, 2049, 41, 3000
    rr = ob.RefRun(rows=synth.treelike_rows(n, m, seed), seed=seed)
    try:
        ctx = api.FitchContext(rr.enc())
        assert rr.getplen(0) == ctx.set_tree(_i32(rr.tree(0)[1]), _i32(rr.tree(0)[2]), rr.root(0))
        _, cl, cr, _, _ = rr.tree(0)
        cands, expect = [], []
        for b in range(B):
            rr.mutate(b % 3)
            _, nl, nr, _, _ = rr.tree(1)
            cands.append(api.edits_between(cl, cr, nl, nr))
            expect.append(rr.getplen(1))
        assert len(set(expect)) > 20                      # a permutation mistake cannot hide behind equal lengths
        # a resident batch is reordered as a whole ...
        resident = ctx.build_batch(cands)
        resident.launch()
        assert np.array_equal(resident.lengths(), np.array(expect))
        resident.free()
        # ... and lvbgpu_score_batch cuts a batch of this size into pieces that are built while the previous one
        # is walked: the pieces' lengths must land where the caller's candidates are
        assert np.array_equal(ctx.score_batch(cands), np.array(expect))
        assert np.array_equal(ctx.score_batch(cands[:2049]), np.array(expect[:2049]))
        ctx.close()
    finally:
        rr.close()


@pytest.mark.parametrize("n,m,seed", [(12, 300, 11), (64, 10000, 12)])
def test_full_batch_matches_reference(api, ob, n, m, seed):
    _ref_or_skip(ob)
    rows = synth.treelike_rows(n, m, seed)
    rr = ob.RefRun(rows=rows, seed=seed)
    try:
        ctx = api.FitchContext(rr.enc())
        lefts, rights, roots, expect = [], [], [], []
        for b in range(16):
            rr.random_tree()
            if b % 4 == 3:
                rr.getplen(0)
                rr.arbreroot()
            expect.append(rr.getplen(0))
            _, l, r, _, _ = rr.tree(0)
            lefts.append(_i32(l))
            rights.append(_i32(r))
            roots.append(rr.root(0))
        got = ctx.score_full_batch(np.stack(lefts), np.stack(rights), roots)
        assert np.array_equal(got, np.array(expect))
        ctx.close()
    finally:
        rr.close()


@pytest.mark.parametrize("n,m,seed", [(10, 40, 21), (100, 984, 22), (40, 5000, 23)])
def test_strict_compat_on_reference_tree_block(api, ob, n, m, seed):
    """lvbgpu_getplen_compat runs on the reference's own BranchArray and must leave it exactly as
    the reference's getplen would (lengths, changes, sets, dirty flags cleared)."""
    _ref_or_skip(ob)
    rows = synth.treelike_rows(n, m, seed)
    ra = ob.RefRun(rows=rows, seed=seed)   # driven by the reference's getplen
    rb = ob.RefRun(rows=rows, seed=seed)   # same RNG stream, driven by the HIP adapter
    try:
        ctx = api.FitchContext(ra.enc())
        assert ctx.getplen_compat(rb.tree_block(0), rb.root(0)) == ra.getplen(0)
        for step in range(40):
            kind = step % 3
            ra.reseed(1000 + step)
            ra.mutate(kind)
            rb.reseed(1000 + step)
            rb.mutate(kind)
            ea = ra.getplen(1)
            eb = ctx.getplen_compat(rb.tree_block(1), rb.root(1))
            assert ea == eb, f"step {step}"
            ta, tb = ra.tree(1), rb.tree(1)
            assert np.array_equal(ta[3][ra.n:], tb[3][rb.n:])
            assert not tb[4][rb.n:].any()
            assert np.array_equal(ra.all_sets(1), rb.all_sets(1))
            if step % 3 == 0:
                ra.swap()
                rb.swap()
            if step % 10 == 9:
                ra.reseed(5000 + step)   # one global RNG inside the reference library:
                ra.arbreroot()           # reseed right before each twin call
                rb.reseed(5000 + step)
                assert rb.arbreroot() == ra.root(0)
                assert ra.getplen(0) == ctx.getplen_compat(rb.tree_block(0), rb.root(0))
        ctx.close()
    finally:
        ra.close()
        rb.close()


def test_device_encoder_matches_reference_encoder(api, ob):
    _ref_or_skip(ob)
    rows = synth.iupac_rows(9, 133, 31)
    rr = ob.RefRun(rows=rows, seed=1)
    try:
        got = api.encode_text(rr.rows())
        assert np.array_equal(got, rr.enc())
    finally:
        rr.close()


def test_device_encoder_rejects_what_the_reference_rejects(api):
    rows = [b"ACGT", b"ACOT", b"ACGT"]  # 'O' passes the reader but DNAToBinary crashes on it
    with pytest.raises(api.LvbGpuError) as ei:
        api.encode_text(rows)
    assert ei.value.status == -7
    assert "bad base symbol" in str(ei.value)


# ------------------------------------------------------------------ committed golden vectors

from tests import goldenlib, helpers  # noqa: E402


@pytest.mark.parametrize("name", goldenlib.names())
def test_device_reproduces_golden_vectors(api, ob, name):
    """tests/golden/vectors/*.npz (from the real reference): encode on the device, full
    evaluations via set_tree, incremental cases via score_batch + commit."""
    g = goldenlib.Golden(name)
    rows = g.rows()
    ctx = api.FitchContext(text_rows=rows)            # device DNAToBinary
    enc = np.stack([ctx.sets(i) for i in range(g.n)])
    assert goldenlib.crc(enc) == int(g.z["enc_crc"])
    resident = None
    for k in range(g.cases):
        c = g.case(k)
        base = int(c["base"])
        if base < 0:
            assert ctx.set_tree(c["left"], c["right"], int(c["root"])) == int(c["length"])
            resident = k
        else:
            b = g.case(base)
            if resident != base:
                ctx.set_tree(b["left"], b["right"], int(b["root"]))
                resident = base
            ed = api.edits_between(b["left"], b["right"], c["left"], c["right"])
            got = ctx.score_batch([ed], roots=[int(c["root"])])[0]
            assert got == int(c["length"]), f"{name} case {k}"
            assert ctx.commit(ed, root=int(c["root"])) == int(c["length"])
            resident = k
        assert np.array_equal(ctx.changes()[g.n:], c["changes"][g.n:])
        if g.nwords <= 1000 or k % 6 == 0:
            assert goldenlib.crc(ctx.all_sets()) == int(c["sets_crc"])
    ctx.close()


def test_rccl_communicator_single_rank(api):
    """lvbgpu_comm_* loads librccl on demand and runs the min-reduce; with one rank the value
    must come back unchanged and the arg-min rank is 0.  (The 8-GPU run belongs to the driver.)"""
    enc = np.full((5, 3), 0x1248124812481248, dtype=np.uint64)
    ctx = api.FitchContext(enc)
    uid = api.comm_unique_id()
    assert len(uid) == 128
    ctx.comm_init(1, 0, uid)
    assert ctx.allreduce_min(123456789012) == (123456789012, 0)
    assert ctx.allreduce_min(7) == (7, 0)
    ctx.close()


def test_wide_offsets_and_copy_path_steps_match_reference(api):
    """fitch_walk has two address forms: 32-bit offsets in 16-byte units (tree blocks below 64 GiB, every
    shape above) and 64-bit byte offsets.  LVBGPU_WIDE_OFFSETS=1 forces the second for the whole process, so
    the same parity cases run once more in ONE child process with it set.  The same child turns direct steps
    and fused commits off (LVBGPU_DIRECT_STEPS=0): the copy / zeroing-launch forms of a step and of a commit
    stay covered as well, and makes the commit walk write its produced sets out in bursts of 3 (LVBGPU_DEFER_SLOTS)
    instead of 32, so bursts end in the middle of chains, and deals up to eight tiles to a wave (LVBGPU_TARGET_WAVES=40:
    by default a launch below 4 M waves walks one tile per wave, so nothing else in the suite groups tiles)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LVBGPU_WIDE_OFFSETS="1", LVBGPU_DIRECT_STEPS="0", LVBGPU_DEFER_SLOTS="3", LVBGPU_TARGET_WAVES="40")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-m", "gpu",
                        "-q", "-x", "-k", "full_evaluation or incremental_batches or full_batch or golden_vectors",
                        "-p", "no:cacheprovider"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], r.stdout[-500:]


def test_two_candidates_per_wave_match_reference(api):
    """LVBGPU_PAIR=1 for a whole child process: every scoring batch - host-built, device-built, of several chains - is walked
    two candidates per wave (fitch_walk_pair; the device-built ones in the order the last workgroups of the generator's own
    launch make).  The parity cases against the compiled reference, the golden vectors and the device-proposal suite (every
    drawn move replayed on the host generators, every device-built program scored against the host-built one) run once more
    that way (the paired walk is optional, so the default runs never exercise it)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LVBGPU_PAIR="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"),
                        os.path.join(root, "tests", "test_gpu_device_proposals.py"), "-m", "gpu", "-q", "-x",
                        "-k", "not wide_offsets and not two_candidates_per_wave", "-p", "no:cacheprovider"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout, r.stdout[-500:]

