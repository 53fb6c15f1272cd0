// api_batch.cpp - liblvbgpu.so: candidates given as edits (or whole topologies) -> programs built on the host -> scored.
#include "ctx.hpp"

extern "C" void lvbgpu_batch_free(lvbgpu_batch *b)
{
    if (b)
    {
        b->d_pairs.release();
        b->h_pairs.release();
    }
    if (!b)
        return;
    if (b->ctx)
    {
        (void)hipSetDevice(b->ctx->device);
        (void)hipStreamSynchronize(b->ctx->stream);
        auto &held = b->ctx->held;
        held.erase(std::remove(held.begin(), held.end(), b), held.end());
    }
    b->d_prog.release();
    b->d_len.release();
    b->h_len.release();
    b->h_stage.release();
    b->h_wflag.release();
    delete b;
}


namespace lvbgpu_detail
{
constexpr int32_t PARALLEL_BUILD_MIN = 256; // below this one thread is as fast (a pool run costs a few microseconds)

// what a batch is made from: edits against the resident tree, or whole topologies
struct BuildJob
{
    const int32_t *edit_offsets = nullptr;
    const lvbgpu_edit *edits = nullptr;
    const int32_t *roots = nullptr;
    const int32_t *left = nullptr, *right = nullptr; // full mode: [B][2n-3]
    bool full = false;
    const int32_t *chain_of = nullptr; // a chain per candidate (lvbgpu_chains_score_edits: every chain's state is parked)
    int32_t par_min = 0;               // candidates from which the pool builds (0: the default)
};

// whole-tree programs of trees [b0, b1)
void build_slice_full(lvbgpu_ctx *ctx, BuildWorker &w, int32_t b0, int32_t b1, const BuildJob &job)
{
    w.topo_version = ~0ull; // the private topology is overwritten below
    w.pb.resize(ctx->nb);
    w.prog.toks.clear();
    w.prog.dsts.clear();
    w.cands.clear();
    w.max_stack = 0;
    w.dirty = 0;
    w.rc = LVBGPU_OK;
    for (int32_t b = b0; b < b1; b++)
    {
        if (!w.topo.assign((int32_t)ctx->n, job.left + (size_t)b * ctx->nb, job.right + (size_t)b * ctx->nb,
                           job.roots ? job.roots[b] : 0, &w.why))
        {
            w.rc = LVBGPU_E_TOPOLOGY;
            w.why = "tree " + std::to_string(b) + ": " + w.why;
            return;
        }
        const size_t tok0 = w.prog.toks.size(), dst0 = w.prog.dsts.size();
        w.prog.max_stack = 0;
        w.pb.build_full(w.topo, w.prog);
        CandDesc cd{};
        cd.tok_off = (uint32_t)tok0;
        cd.ntok = (uint32_t)(w.prog.toks.size() - tok0);
        cd.dst_off = (uint32_t)dst0;
        cd.ncomb = (uint32_t)(w.prog.dsts.size() - dst0);
        for (size_t i = tok0; i < w.prog.toks.size(); i++)
            cd.nfresh += (w.prog.toks[i] & TOK_FRESH) ? 1u : 0u;
        w.cands.push_back(cd);
        w.max_stack = std::max(w.max_stack, w.prog.max_stack);
        w.dirty += w.prog.dirty;
    }
}

// programs of candidates [b0, b1) with one worker's private topology copy
void build_slice(lvbgpu_ctx *ctx, BuildWorker &w, int32_t b0, int32_t b1, const BuildJob &job)
{
    if (job.full)
        return build_slice_full(ctx, w, b0, b1, job);
    const int32_t *edit_offsets = job.edit_offsets;
    const lvbgpu_edit *edits = job.edits;
    const int32_t *roots = job.roots;
    if (!job.chain_of && w.topo_version != ctx->topo_version)
    {
        w.topo = ctx->topo;
        w.topo_version = ctx->topo_version;
        w.pb.resize(w.topo.nb);
    }
    w.prog.toks.clear();
    w.prog.dsts.clear();
    w.cands.clear();
    w.max_stack = 0;
    w.dirty = 0;
    w.rc = LVBGPU_OK;
    for (int32_t b = b0; b < b1; b++)
    {
        const int32_t e0 = edit_offsets[b], e1 = edit_offsets[b + 1];
        if (e1 < e0)
        {
            w.rc = LVBGPU_E_ARG;
            w.why = "edit_offsets not monotone";
            return;
        }
        int32_t chain = ctx->chain;
        if (job.chain_of) // the candidate's own chain: this worker's private copy of that chain's topology
        {
            chain = job.chain_of[b];
            if (chain < 0 || chain >= ctx->nchains || !ctx->parked[(size_t)chain].have_tree)
            {
                w.rc = chain < 0 || chain >= ctx->nchains ? LVBGPU_E_ARG : LVBGPU_E_STATE;
                w.why = "candidate " + std::to_string(b) + ": chain out of range, or it has no resident tree";
                return;
            }
            const ChainSlot &cs = ctx->parked[(size_t)chain];
            if (w.topo_version != cs.topo_version)
            {
                w.topo = cs.topo;
                w.topo_version = cs.topo_version;
                w.pb.resize(w.topo.nb);
            }
        }
        const size_t tok0 = w.prog.toks.size(), dst0 = w.prog.dsts.size();
        w.prog.max_stack = 0;
        if (!w.pb.build_candidate(w.topo, reinterpret_cast<const Edit *>(edits) + e0, e1 - e0, roots ? roots[b] : -1,
                                  w.prog, &w.why))
        {
            w.rc = LVBGPU_E_TOPOLOGY;
            w.why = "candidate " + std::to_string(b) + ": " + w.why;
            return;
        }
        CandDesc cd{};
        cd.tok_off = (uint32_t)tok0;
        cd.ntok = (uint32_t)(w.prog.toks.size() - tok0);
        cd.dst_off = (uint32_t)dst0;
        cd.ncomb = (uint32_t)(w.prog.dsts.size() - dst0);
        cd.flags = CAND_RESIDENT_BASE | ((uint32_t)chain << CAND_CHAIN_SHIFT);
        for (size_t i = tok0; i < w.prog.toks.size(); i++)
            cd.nfresh += (w.prog.toks[i] & TOK_FRESH) ? 1u : 0u;
        w.cands.push_back(cd);
        w.max_stack = std::max(w.max_stack, w.prog.max_stack);
        w.dirty += w.prog.dirty;
    }
}

// fill `bt` (new or recycled: its buffers only ever grow) with the programs of B candidates:
// slices of the batch are built on the pool's threads straight into the pinned upload buffer
int build_into(lvbgpu_ctx *ctx, lvbgpu_batch *bt, int32_t B, const BuildJob &job)
{
    static_assert(sizeof(lvbgpu_edit) == sizeof(Edit), "edit layout");
    int T = 1;
    // whole-tree programs are ~n tokens each: worth the pool from a handful of trees on
    static const int32_t par_min_edits = [] {
        const char *e = getenv("LVBGPU_PAR_MIN"); // candidates from which a build uses the pool (two per thread at least)
        const int v = e ? atoi(e) : PARALLEL_BUILD_MIN;
        return (int32_t)(v < 2 ? 2 : v);
    }();
    const int32_t par_min = job.par_min > 0 ? job.par_min : (job.full ? 16 : par_min_edits);
    if (B >= par_min)
    {
        if (!ctx->pool)
        {
            const int n = host_threads();
            if (n > 1)
                ctx->pool = new (std::nothrow) Pool(n);
        }
        if (ctx->pool)
            T = std::max(1, std::min(ctx->pool->size(), B / (par_min / 2)));
    }
    if ((int)ctx->workers.size() < T)
        ctx->workers.resize(T);
    auto slice = [&](int t) {
        build_slice(ctx, ctx->workers[t], (int32_t)((int64_t)B * t / T), (int32_t)((int64_t)B * (t + 1) / T), job);
    };
    if (T == 1)
        slice(0);
    else
        ctx->pool->run(T, slice);

    size_t ntok = 0, ndst = 0;
    std::vector<size_t> tok_base(T), dst_base(T), cand_base(T);
    int32_t max_stack = 0;
    int64_t dirty = 0;
    size_t ncand = 0;
    for (int t = 0; t < T; t++)
    {
        BuildWorker &w = ctx->workers[t];
        if (w.rc != LVBGPU_OK)
            return ctx->fail(w.rc, w.why);
        tok_base[t] = ntok;
        dst_base[t] = ndst;
        cand_base[t] = ncand;
        ntok += w.prog.toks.size();
        ndst += w.prog.dsts.size();
        ncand += w.cands.size();
        max_stack = std::max(max_stack, w.max_stack);
        dirty += w.dirty;
    }
    int rc = check_depth(ctx, max_stack);
    if (rc != LVBGPU_OK)
        return rc;
    if ((uint64_t)B * ctx->ntiles >= (1ull << 31) || ntok >= (1ull << 32))
        return ctx->fail(LVBGPU_E_ARG, "batch too large: B * tiles must stay below 2^31");

    const size_t o_t = align16((size_t)B * sizeof(CandDesc));
    const size_t o_d = o_t + align16(ntok * 4);
    const size_t total = o_d + align16(ndst * 4);
    HIPCHK(ctx, bt->d_prog.reserve(total));
    // staging: a recycled step batch owns its pinned buffer (several may be in flight, lvbgpu_score_batch);
    // a batch the caller holds goes through the context's and waits for the copy below
    PinBuf &stage = bt->recycled ? bt->h_stage : ctx->h_pin;
    HIPCHK(ctx, stage.reserve(total));
    char *h = (char *)stage.p;
    // big launches: within each run of `chunk` candidates, longest programs first.  The chip runs ~12 rounds of
    // waves per launch at B = 4096 and a wave's time follows its token count (3 .. 60+), so whatever runs last
    // decides the tail: let that be short ones.  Chunks (an eighth of the batch), not one global order: every XCD
    // takes a contiguous slice of the candidate list and must get the same mix of long and short (a global order
    // cost 7 % at B = 16 384).  Candidate b's descriptor (and so its length slot) moves to position slot_of[b];
    // lvbgpu_batch_lengths undoes it.
    bt->slot_of.clear();
    // two candidates per wave (fitch_walk_pair): candidates sorted by their program READ BACKWARDS - neighbours in that
    // order share the longest suffixes - and taken two by two; programs of more than one 64-token chunk walk alone.  The
    // pairs replace the longest-first layout (a wave's time is its two programs' private parts plus the shared one).
    const bool pair_up = ctx->pair_min > 0 && B >= ctx->pair_min && !job.full;
    bt->npairs = 0;
    std::vector<uint32_t> pair_list;
    if (pair_up)
    {
        std::vector<const uint32_t *> tokp((size_t)B);
        std::vector<uint32_t> ntok_of((size_t)B);
        {
            size_t b = 0;
            for (int t = 0; t < T; t++)
                for (const CandDesc &c : ctx->workers[t].cands)
                {
                    tokp[b] = ctx->workers[t].prog.toks.data() + c.tok_off;
                    ntok_of[b++] = c.ntok;
                }
        }
        std::vector<uint32_t> shorts, longs;
        for (uint32_t b = 0; b < (uint32_t)B; b++)
            (ntok_of[b] >= 1u && ntok_of[b] <= 64u ? shorts : longs).push_back(b);
        std::sort(shorts.begin(), shorts.end(), [&](uint32_t x, uint32_t y) {
            const uint32_t nx = ntok_of[x], ny = ntok_of[y];
            for (uint32_t i = 1; i <= nx && i <= ny; i++)
                if (tokp[x][nx - i] != tokp[y][ny - i])
                    return tokp[x][nx - i] < tokp[y][ny - i];
            return nx != ny ? nx < ny : x < y;
        });
        for (size_t i = 0; i + 1 < shorts.size(); i += 2)
        {
            pair_list.push_back(shorts[i]);
            pair_list.push_back(shorts[i + 1]);
        }
        if (shorts.size() & 1u)
        {
            pair_list.push_back(shorts.back());
            pair_list.push_back(PICK_NONE);
        }
        for (uint32_t b : longs)
        {
            pair_list.push_back(b);
            pair_list.push_back(PICK_NONE);
        }
        bt->npairs = (uint32_t)(pair_list.size() / 2);
    }
    if (B >= LPT_MIN_B && ctx->lpt_order && !pair_up)
    {
        std::vector<uint32_t> ntok_of((size_t)B);
        {
            size_t b = 0;
            for (int t = 0; t < T; t++)
                for (const CandDesc &c : ctx->workers[t].cands)
                    ntok_of[b++] = c.ntok;
        }
        static const int32_t nchunks = [] {
            const char *e = getenv("LVBGPU_LPT_CHUNKS");
            const int v = e ? atoi(e) : 8;
            return v < 1 ? 1 : v;
        }();
        const int32_t chunk = std::max(64, B / nchunks);
        bt->slot_of.resize((size_t)B);
        std::vector<uint32_t> start;
        for (int32_t c0 = 0; c0 < B; c0 += chunk)
        {
            const int32_t c1 = std::min(B, c0 + chunk);
            uint32_t max_tok = 0;
            for (int32_t b = c0; b < c1; b++)
                max_tok = std::max(max_tok, ntok_of[(size_t)b]);
            start.assign((size_t)max_tok + 2, 0); // counting sort, descending token count, stable
            for (int32_t b = c0; b < c1; b++)
                start[max_tok - ntok_of[(size_t)b] + 1]++;
            for (size_t i = 1; i < start.size(); i++)
                start[i] += start[i - 1];
            for (int32_t b = c0; b < c1; b++)
                bt->slot_of[(size_t)b] = c0 + (int32_t)start[max_tok - ntok_of[(size_t)b]]++;
        }
    }
    auto gather = [&](int t) {
        BuildWorker &w = ctx->workers[t];
        CandDesc *cd = (CandDesc *)h;
        for (size_t i = 0; i < w.cands.size(); i++)
        {
            const size_t b = cand_base[t] + i;
            CandDesc &out = cd[bt->slot_of.empty() ? b : (size_t)bt->slot_of[b]];
            out = w.cands[i];
            out.tok_off += (uint32_t)tok_base[t];
            out.dst_off += (uint32_t)dst_base[t];
        }
        memcpy(h + o_t + tok_base[t] * 4, w.prog.toks.data(), w.prog.toks.size() * 4);
        memcpy(h + o_d + dst_base[t] * 4, w.prog.dsts.data(), w.prog.dsts.size() * 4);
    };
    if (T == 1)
        gather(0);
    else
        ctx->pool->run(T, gather);
    // small recycled steps: lengths return through the walk's last wave, and a small enough batch of
    // programs is read where it lies (pinned, device-visible) instead of being copied first
    const uint32_t ngroups = choose_groups((uint32_t)B, ctx->ntiles, ctx->target_waves);
    bt->direct = bt->recycled && ctx->direct_steps && (uint64_t)B * ngroups <= DIRECT_STEP_MAX_ITEMS && !pair_up;
    bt->in_place = bt->direct && total * ngroups <= DIRECT_READ_MAX_BYTES;
    if (!bt->in_place)
        HIPCHK(ctx, hipMemcpyAsync(bt->d_prog.p, h, total, hipMemcpyHostToDevice, ctx->stream));
    if (bt->npairs)
    {
        const size_t pbytes = pair_list.size() * 4;
        if (pbytes > bt->h_pairs.cap) // (the pinned copy the upload reads: grown only when the stream holds nothing of it)
        {
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            HIPCHK(ctx, bt->h_pairs.reserve(pbytes));
        }
        HIPCHK(ctx, bt->d_pairs.reserve(pbytes));
        memcpy(bt->h_pairs.p, pair_list.data(), pbytes);
        HIPCHK(ctx, hipMemcpyAsync(bt->d_pairs.p, bt->h_pairs.p, pbytes, hipMemcpyHostToDevice, ctx->stream));
    }
    // h_pin is reused by the next upload.  A recycled step batch is read back (and the stream
    // drained) by lvbgpu_batch_lengths before anything can build again: no need to wait here.
    if (!bt->recycled)
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    bt->ctx = ctx;
    bt->B = B;
    bt->off_toks = o_t;
    bt->off_dsts = o_d;
    // a buffer that grew holds whatever its new memory held - and may well sit at the old address, so it is
    // the capacity that tells, not the pointer
    const size_t old_len_cap = bt->d_len.cap;
    HIPCHK(ctx, bt->d_len.reserve((size_t)B * 8));
    if (bt->d_len.cap != old_len_cap)
        bt->len_zeroed = false;
    HIPCHK(ctx, bt->h_len.reserve((size_t)B * 8));
    bt->full_mode = job.full;
    bt->topo_version = ctx->topo_version;
    bt->chain = ctx->chain;
    bt->spans_chains = job.chain_of != nullptr;
    bt->build_gen++;
    bt->stats.candidates = B;
    bt->stats.combines = (int64_t)ndst;
    bt->stats.rows_read = (int64_t)ntok;
    bt->stats.dirty_nodes = dirty;
    bt->stats.max_stack = max_stack;
    bt->stats.algorithmic_bytes = bt->stats.rows_read * ctx->nwords * 8;
    return LVBGPU_OK;
}
} // namespace lvbgpu_detail

extern "C" int lvbgpu_batch_build(lvbgpu_ctx *ctx, int32_t B, const int32_t *edit_offsets, const lvbgpu_edit *edits,
                                  const int32_t *roots, lvbgpu_batch **out)
{
    if (!ctx || !out || B < 1 || !edit_offsets || (!edits && edit_offsets[B] > 0))
        return LVBGPU_E_ARG;
    {
        const int rf = settle(ctx); // the host side of a chain commit still on its way
        if (rf != LVBGPU_OK)
            return rf;
    }
    *out = nullptr;
    if (!ctx->have_tree)
        return ctx->fail(LVBGPU_E_STATE, "no resident tree: call lvbgpu_set_tree first");
    ENTER(ctx);
    lvbgpu_batch *bt = new (std::nothrow) lvbgpu_batch();
    if (!bt)
        return LVBGPU_E_NOMEM;
    BuildJob job;
    job.edit_offsets = edit_offsets;
    job.edits = edits;
    job.roots = roots;
    const int rc = build_into(ctx, bt, B, job);
    if (rc != LVBGPU_OK)
    {
        lvbgpu_batch_free(bt);
        return rc;
    }
    ctx->held.push_back(bt);
    *out = bt;
    return LVBGPU_OK;
}

extern "C" int lvbgpu_batch_launch(lvbgpu_ctx *ctx, lvbgpu_batch *b)
{
    if (!ctx || !b || b->ctx != ctx)
        return LVBGPU_E_ARG;
    // candidates are rewrites of ONE resident tree: a batch kept across a commit would be scored against rows
    // that no longer mean what its programs assume (whole-topology batches read leaf rows only and stay valid)
    if (!b->full_mode && !b->spans_chains && (b->chain >= ctx->nchains || b->topo_version != ctx->version_of(b->chain)))
        return ctx->fail(LVBGPU_E_STATE, "the resident tree changed since this batch was built");
    ENTER(ctx);
    if (!b->len_zeroed) // the whole buffer: a direct step's last wave re-zeroes only the B slots it used
        HIPCHK(ctx, hipMemsetAsync(b->d_len.p, 0, b->recycled ? b->d_len.cap : (size_t)b->B * 8, ctx->stream));
    b->len_zeroed = false;
    WalkArgs a = resident_args(ctx, b->in_place ? b->h_stage.p : b->d_prog.p, b->off_toks, b->off_dsts, b->d_len.p,
                               (uint32_t)b->B, (int32_t)b->stats.max_stack);
    if (b->direct)
    {
        a.host_len = (unsigned long long *)b->h_len.p;
        a.done_count = (uint32_t *)(ctx->d_scalars + 2);
        a.host_flag = (uint32_t *)ctx->h_step.p;
        a.step_seq = ++ctx->step_seq;
    }
    else if (b->watch_flag)
    {
        a.host_len = (unsigned long long *)b->h_len.p;
        a.host_flag = b->watch_flag;
        a.step_seq = b->watch_seq;
        a.watcher = 1; // (launch_walk turns it off again for launches it does not fit)
    }
    b->own_watch = false;
    if (!b->direct && !b->watch_flag && b->recycled && !b->full_mode)
    {
        static const bool allow_watcher = [] {
            const char *e = getenv("LVBGPU_WATCHER");
            return !(e && e[0] == '0');
        }();
        if (allow_watcher && a.ngroups <= WATCH_MAX_GROUPS && b->B <= 8192)
        {
            if (!b->h_wflag.p)
            {
                HIPCHK(ctx, b->h_wflag.reserve(WATCH_WAVES * 4));
                memset(b->h_wflag.p, 0, WATCH_WAVES * 4);
            }
            b->own_seq = ++b->own_seq == 0xFFFFFFFFu ? 1u : b->own_seq; // (0xFFFFFFFF is the watcher's "gave up")
            a.host_len = (unsigned long long *)b->h_len.p;
            a.host_flag = (uint32_t *)b->h_wflag.p;
            a.step_seq = b->own_seq;
            a.watcher = 1;
            b->own_watch = true;
        }
    }
    a.watch_starve = ctx->starve_watcher ? 1u : 0u;
    if (b->npairs && !b->direct)
    {
        a.pairs = (const uint32_t *)b->d_pairs.p;
        a.npairs = b->npairs;
        a.nitems = b->npairs * a.ngroups;
        ctx->paired_walks++;
    }
    const bool timed = ctx->walk_timing && (ctx->wt_seen++ % ctx->wt_every) == 0;
    if (timed)
    {
        if (ctx->wt_pending == lvbgpu_ctx::WT_RING)
        {
            const int rc = walk_timing_drain(ctx);
            if (rc != LVBGPU_OK)
                return rc;
        }
        HIPCHK(ctx, hipEventRecord(ctx->wt_ev[2 * ctx->wt_pending], ctx->stream));
    }
    HIPCHK(ctx, launch_walk(a, false, ctx->stream, &ctx->flip_counter));
    if (timed)
    {
        HIPCHK(ctx, hipEventRecord(ctx->wt_ev[2 * ctx->wt_pending + 1], ctx->stream));
        ctx->wt_pending++;
    }
    b->launched = true;
    return LVBGPU_OK;
}

namespace lvbgpu_detail
{
// a direct step's lengths are in h_len once the walk's last wave has released the step's sequence number
static hipError_t wait_for_direct_step(lvbgpu_ctx *ctx)
{
    const uint32_t *flag = (const uint32_t *)ctx->h_step.p;
    const WaitClock clock(ctx->wait_limit_s);
    for (uint32_t spins = 1;; spins++) // (memory only: hipStreamQuery can block behind a running kernel, api_propose.cpp)
    {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == ctx->step_seq)
            return hipSuccess;
        if ((spins & 1023u) == 0 && clock.expired())
        {
            ctx->wait_gave_up = true;
            return hipErrorNotReady; // (fail_hip words it: the wait limit)
        }
    }
}
} // namespace lvbgpu_detail

extern "C" int lvbgpu_batch_lengths(lvbgpu_ctx *ctx, lvbgpu_batch *b, int64_t *lengths_out)
{
    if (!ctx || !b || b->ctx != ctx || !lengths_out)
        return LVBGPU_E_ARG;
    if (!b->launched)
        return ctx->fail(LVBGPU_E_STATE, "this batch was never launched: call lvbgpu_batch_launch first");
    ENTER(ctx);
    if (b->direct)
    {
        HIPCHK(ctx, wait_for_direct_step(ctx));
        b->len_zeroed = true; // by the walk's last wave
    }
    else if (b->own_watch)
    {
        // the walk's watcher waves store the lengths into h_len and set their words: nothing but memory is polled
        const uint32_t *flag = (const uint32_t *)b->h_wflag.p;
        const WaitClock clock(ctx->wait_limit_s);
        uint32_t seen = b->own_seq, w = 0;
        for (uint32_t spins = 1; w < WATCH_WAVES; spins++)
        {
            seen = __atomic_load_n(flag + w, __ATOMIC_ACQUIRE);
            if (seen == b->own_seq)
            {
                w++;
                continue;
            }
            if (seen == 0xFFFFFFFFu)
                break;
            if ((spins & 4095u) == 0 && clock.expired())
            {
                ctx->wait_gave_up = true;
                return ctx->fail_wait("lvbgpu_batch_lengths: the walk's watcher waves", clock.waited());
            }
        }
        if (seen != b->own_seq)
            return ctx->fail(LVBGPU_E_STATE, "the walk's watcher did not hand the lengths over (flag " + std::to_string(seen) + ")");
    }
    else
    {
        // through pinned memory: one DMA, no staging
        HIPCHK(ctx, hipMemcpyAsync(b->h_len.p, b->d_len.p, (size_t)b->B * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (b->recycled)
            HIPCHK(ctx, wait_for_step(ctx, b->B));
        else
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (b->recycled && !b->direct)
    {
        // clear the slots for the next step now, while the host consumes these lengths
        HIPCHK(ctx, hipMemsetAsync(b->d_len.p, 0, b->d_len.cap, ctx->stream));
        b->len_zeroed = true;
    }
    if (b->slot_of.empty())
        memcpy(lengths_out, b->h_len.p, (size_t)b->B * 8);
    else // candidates were laid out longest program first
        for (int32_t i = 0; i < b->B; i++)
            lengths_out[i] = ((const int64_t *)b->h_len.p)[b->slot_of[(size_t)i]];
    for (int32_t i = 0; i < b->B; i++)
        if (lengths_out[i] <= 0)
            return ctx->fail(LVBGPU_E_ZEROLEN, "assertion failed: changes > 0 (candidate " + std::to_string(i) + " of " +
                                                   std::to_string(b->B) + " scored " + std::to_string(lengths_out[i]) + ")");
    return LVBGPU_OK;
}

extern "C" int lvbgpu_batch_get_stats(const lvbgpu_batch *b, lvbgpu_batch_stats *out)
{
    if (!b || !out)
        return LVBGPU_E_ARG;
    *out = b->stats;
    return LVBGPU_OK;
}

extern "C" int lvbgpu_score_batch(lvbgpu_ctx *ctx, int32_t B, const int32_t *edit_offsets, const lvbgpu_edit *edits,
                                  const int32_t *roots, int64_t *lengths_out)
{
    if (!ctx || !lengths_out || B < 1 || !edit_offsets || (!edits && edit_offsets[B] > 0))
        return LVBGPU_E_ARG;
    {
        const int rf = settle(ctx); // the host side of a chain commit still on its way
        if (rf != LVBGPU_OK)
            return rf;
    }
    if (!ctx->have_tree)
        return ctx->fail(LVBGPU_E_STATE, "no resident tree: call lvbgpu_set_tree first");
    ENTER(ctx);
    // recycled batches owned by the context: a search calls this every step, so no allocation here.  A big batch
    // is cut into STEP_PIPELINE pieces: while the device walks one piece the host threads build the next, so the
    // device time of all but the last piece hides behind program building (the larger part of such a step).
    // pieces of at least STEP_PIPELINE_PIECE candidates: smaller ones do not keep all host threads busy
    const int K = ctx->pipeline_steps ? std::max(1, std::min(lvbgpu_ctx::STEP_PIPELINE, B / STEP_PIPELINE_PIECE)) : 1;
    int rc = LVBGPU_OK;
    for (int k = 0; k < K && rc == LVBGPU_OK; k++)
    {
        lvbgpu_batch *&b = ctx->step_batch[k];
        if (!b)
        {
            b = new (std::nothrow) lvbgpu_batch();
            if (!b)
                return LVBGPU_E_NOMEM;
            b->recycled = true;
        }
        const int32_t c0 = (int32_t)((int64_t)B * k / K), c1 = (int32_t)((int64_t)B * (k + 1) / K);
        BuildJob job;
        job.edit_offsets = edit_offsets + c0; // offsets stay absolute into `edits`
        job.edits = edits;
        job.roots = roots ? roots + c0 : nullptr;
        rc = build_into(ctx, b, c1 - c0, job);
        if (rc == LVBGPU_OK)
            rc = lvbgpu_batch_launch(ctx, b);
        else if (rc == LVBGPU_E_TOPOLOGY || rc == LVBGPU_E_ARG)
            ctx->last_error += " (piece starting at candidate " + std::to_string(c0) + ")";
    }
    for (int k = 0; k < K && rc == LVBGPU_OK; k++)
        rc = lvbgpu_batch_lengths(ctx, ctx->step_batch[k], lengths_out + (int64_t)B * k / K);
    if (rc != LVBGPU_OK)
    {
        // Leave nothing in flight that reads the staging buffers.  After an ordinary failure (bad edits in a later
        // piece) the stream is healthy: drain it.  After a wait that gave up it may be stuck, and draining would be the
        // endless wait the limit exists to prevent: the step batches - whose buffers a queued walk may still read and
        // write - are set aside instead (freed with the context, after its streams have been waited for) and the next
        // call makes new ones.
        if (ctx->wait_gave_up)
        {
            ctx->wait_gave_up = false;
            for (lvbgpu_batch *&b : ctx->step_batch)
                if (b)
                {
                    ctx->set_aside.push_back(b);
                    b = nullptr;
                }
        }
        else
            (void)hipStreamSynchronize(ctx->stream);
    }
    return rc;
}

namespace lvbgpu_detail
{
void park_chain(lvbgpu_ctx *ctx);
void unpark_chain(lvbgpu_ctx *ctx, int32_t c);
uint64_t edits_hash(const lvbgpu_edit *e, int32_t n)
{
    uint64_t h = 0xCBF29CE484222325ull; // FNV-1a over the rewrites as given
    const unsigned char *p = reinterpret_cast<const unsigned char *>(e);
    for (size_t i = 0; i < (size_t)n * sizeof(lvbgpu_edit); i++)
        h = (h ^ p[i]) * 0x100000001B3ull;
    return h;
}
} // namespace lvbgpu_detail

// fn(i, arg) for i in [0, n) on the context's host threads (the pool that builds programs), the caller among them;
// returns when all are done.  For host loops that prepare many chains' candidates between two library calls.
extern "C" int lvbgpu_parallel_for(lvbgpu_ctx *ctx, int32_t n, lvbgpu_task_fn fn, void *arg)
{
    if (!ctx || n < 0 || !fn)
        return LVBGPU_E_ARG;
    if (n > 1 && !ctx->pool)
    {
        const int t = host_threads();
        if (t > 1)
            ctx->pool = new (std::nothrow) Pool(t);
    }
    const int T = ctx->pool ? std::max(1, std::min(ctx->pool->size(), (int)n)) : 1;
    if (T == 1)
    {
        for (int32_t i = 0; i < n; i++)
            fn(i, arg);
        return LVBGPU_OK;
    }
    ctx->pool->run(T, [&](int t) {
        for (int32_t i = (int32_t)((int64_t)n * t / T), e = (int32_t)((int64_t)n * (t + 1) / T); i < e; i++)
            fn(i, arg);
    });
    return LVBGPU_OK;
}

// Host-made candidates of SEVERAL chains in one walk: candidate b is a set of rewrites of chain chain_of[b]'s resident
// tree (any set that gives a tree: one move, or the cumulative rewrites of a run of moves).  What lvbgpu_select_chain +
// lvbgpu_score_batch does chain by chain, with one launch for all of them; the programs are built on the pool's threads
// from 32 candidates on.  Candidates of one chain should be adjacent (a builder thread keeps one chain's topology at a time).
extern "C" int lvbgpu_chains_score_edits(lvbgpu_ctx *ctx, int32_t B, const int32_t *chain_of, const int32_t *edit_offsets,
                                         const lvbgpu_edit *edits, int64_t *lengths_out)
{
    if (!ctx || !lengths_out || B < 1 || !chain_of || !edit_offsets || (!edits && edit_offsets[B] > 0))
        return LVBGPU_E_ARG;
    {
        const int rf = settle(ctx); // the host side of a chain commit still on its way
        if (rf != LVBGPU_OK)
            return rf;
    }
    ENTER(ctx);
    struct Parked
    {
        lvbgpu_ctx *c;
        int32_t sel;
        explicit Parked(lvbgpu_ctx *x) : c(x), sel(x->chain) { park_chain(x); }
        ~Parked() { unpark_chain(c, sel); }
    } guard(ctx);
    lvbgpu_batch *&b = ctx->step_batch[0];
    if (!b)
    {
        b = new (std::nothrow) lvbgpu_batch();
        if (!b)
            return LVBGPU_E_NOMEM;
        b->recycled = true;
    }
    BuildJob job;
    job.edit_offsets = edit_offsets;
    job.edits = edits;
    job.chain_of = chain_of;
    job.par_min = 32;
    for (BuildWorker &w : ctx->workers)
        w.topo_version = ~0ull; // (versions are unique across chains, but a worker's copy may be of a chain re-uploaded since)
    ctx->scored.clear();
    ctx->scored_batch = nullptr;
    int rc = build_into(ctx, b, B, job);
    if (rc == LVBGPU_OK)
        rc = lvbgpu_batch_launch(ctx, b);
    if (rc == LVBGPU_OK)
        rc = lvbgpu_batch_lengths(ctx, b, lengths_out);
    if (rc != LVBGPU_OK)
    {
        if (ctx->wait_gave_up)
        {
            ctx->wait_gave_up = false;
            ctx->set_aside.push_back(b);
            b = nullptr;
        }
        else
            (void)hipStreamSynchronize(ctx->stream);
        return rc;
    }
    // what was scored, for the commit that follows (programs read in place from pinned memory are not kept: the next
    // build overwrites them while a commit walk might still read)
    if (!b->in_place && b->slot_of.empty())
    {
        ctx->scored.resize((size_t)B);
        ctx->scored_edits.assign(edits, edits + edit_offsets[B]); // (the commit compares the rewrites themselves on a hash hit)
        for (int32_t i = 0; i < B; i++)
        {
            const int32_t n = edit_offsets[i + 1] - edit_offsets[i];
            ctx->scored[(size_t)i] = {chain_of[i], n, edit_offsets[i], ctx->parked[(size_t)chain_of[i]].topo_version,
                                      edits_hash(edits + edit_offsets[i], n)};
        }
        ctx->scored_batch = b;
        ctx->scored_gen = b->build_gen;
    }
    return rc;
}

extern "C" int lvbgpu_score_full_batch(lvbgpu_ctx *ctx, int32_t B, const int32_t *left, const int32_t *right,
                                       const int32_t *roots, int64_t *lengths_out)
{
    if (!ctx || B < 1 || !left || !right || !lengths_out)
        return LVBGPU_E_ARG;
    ENTER(ctx);
    if (!ctx->full_batch)
    {
        ctx->full_batch = new (std::nothrow) lvbgpu_batch();
        if (!ctx->full_batch)
            return LVBGPU_E_NOMEM;
        ctx->full_batch->recycled = true;
    }
    lvbgpu_batch *bt = ctx->full_batch; // recycled like the step batch
    BuildJob job;
    job.left = left;
    job.right = right;
    job.roots = roots;
    job.full = true;
    int rc = build_into(ctx, bt, B, job);
    if (rc == LVBGPU_OK)
        rc = lvbgpu_batch_launch(ctx, bt);
    if (rc == LVBGPU_OK)
        rc = lvbgpu_batch_lengths(ctx, bt, lengths_out);
    return rc;
}

