// api_propose.cpp - liblvbgpu.so: neighbourhoods whose rewrites and programs are built on the device (drawn there, or
// named by the host) - for one resident tree, or for several chains in ONE generator launch and ONE walk.
#include "ctx.hpp"
#include "gather.hpp"

namespace lvbgpu_detail
{
void park_chain(lvbgpu_ctx *ctx);
void unpark_chain(lvbgpu_ctx *ctx, int32_t c);

// Multi-chain calls work on lvbgpu_ctx::parked only: the selected chain's state is put there on entry and taken
// back on every way out.
struct AllParked
{
    lvbgpu_ctx *ctx;
    int32_t sel;
    explicit AllParked(lvbgpu_ctx *c) : ctx(c), sel(c->chain) { park_chain(c); }
    ~AllParked() { unpark_chain(ctx, sel); }
    AllParked(const AllParked &) = delete;
    AllParked &operator=(const AllParked &) = delete;
};

// the generator's tables of one topology (layout: GenArgs in kernels.hpp), written where the upload reads them
template <typename IdxT>
static uint32_t fill_tables(const Topology &t, IdxT *out, size_t cap_elems, int32_t K)
{
    const size_t nb = (size_t)t.nb, nlo = (size_t)t.n;
    // preorder from the root leaf; children before parents when read backwards
    std::vector<int32_t> order, depth(nb, 0), nleaf(nb, 1), st{t.root};
    order.reserve(nb);
    while (!st.empty())
    {
        const int32_t v = st.back();
        st.pop_back();
        order.push_back(v);
        if (t.left[v] >= 0)
        {
            depth[t.left[v]] = depth[t.right[v]] = depth[v] + 1;
            st.push_back(t.right[v]);
            st.push_back(t.left[v]);
        }
    }
    for (auto it = order.rbegin(); it != order.rend(); ++it)
        if (t.left[*it] >= 0)
            nleaf[*it] = nleaf[t.left[*it]] + nleaf[t.right[*it]];
    // K is the same for every tree of these taxa (2^K exceeds any depth): the device rebuilds a slot in place
    const size_t elems = (7 + (size_t)K) * nb + nlo;
    if (elems + 8 > cap_elems)
        return 0;
    memset(out, 0, (elems + 8) * sizeof(IdxT));
    IdxT *parent = out, *left = parent + nb, *right = left + nb, *nl = right + nb, *dep = nl + nb, *tin = dep + nb,
         *first = tin + nb, *lo = first + nb, *up = lo + nlo;
    size_t nleaves = 0;
    for (size_t i = 0; i < order.size(); i++)
    {
        const int32_t v = order[i];
        tin[v] = (IdxT)i;
        first[v] = (IdxT)nleaves;
        if (t.left[v] < 0)
            lo[nleaves++] = (IdxT)v;
    }
    for (size_t v = 0; v < nb; v++)
    {
        parent[v] = (IdxT)((int32_t)v == t.root ? t.root : t.parent[v]);
        left[v] = (IdxT)(t.left[v] < 0 ? 0 : t.left[v]);
        right[v] = (IdxT)(t.right[v] < 0 ? 0 : t.right[v]);
        nl[v] = (IdxT)nleaf[v];
        dep[v] = (IdxT)depth[v];
        up[v] = parent[v];
    }
    for (int32_t k = 1; k < K; k++)
        for (size_t v = 0; v < nb; v++)
            up[(size_t)k * nb + v] = up[(size_t)(k - 1) * nb + (size_t)up[(size_t)(k - 1) * nb + v]];
    return (uint32_t)(((elems + 8) * sizeof(IdxT)) & ~(size_t)15); // whole 16-byte pieces
}

// bring the device tables of the listed chains (all parked) up to date: the stale ones are rebuilt on the host
// threads, each into its own pinned staging slot, and uploaded.  A staging slot is rewritten only after the step
// that followed its last upload has been waited for, so no upload can still be reading it.
int prepare_tables(lvbgpu_ctx *ctx, const int32_t *chains, int32_t k, bool *uploaded = nullptr)
{
    if (uploaded)
        *uploaded = false;
    const int32_t nb = ctx->nb;
    ctx->gen_idx_bytes = nb <= 65535 ? 2u : 4u;
    int32_t kmax = 1;
    while ((1 << kmax) <= nb)
        kmax++;
    ctx->gen_kmax = kmax;
    {
        const size_t widest = ((7 + (size_t)kmax) * (size_t)nb + (size_t)ctx->n + 8) * ctx->gen_idx_bytes;
        ctx->gen_table_stride = (uint32_t)((widest + 255) & ~(size_t)255);
    }
    const size_t old_cap = ctx->d_topo4.cap;
    HIPCHK(ctx, ctx->d_topo4.reserve((size_t)ctx->nchains * ctx->gen_table_stride));
    if (ctx->d_topo4.cap != old_cap) // a new buffer holds no chain's tables
        for (ChainSlot &cs : ctx->parked)
            cs.d_topo_version = ~0ull;
    const size_t old_stage = ctx->h_topo.cap;
    if ((size_t)ctx->nchains * ctx->gen_table_stride > old_stage)
    {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); // nothing may still read the buffer that is about to go
        HIPCHK(ctx, ctx->h_topo.reserve((size_t)ctx->nchains * ctx->gen_table_stride));
    }
    std::vector<int32_t> stale;
    for (int32_t i = 0; i < k; i++)
        if (ctx->parked[(size_t)chains[i]].d_topo_version != ctx->parked[(size_t)chains[i]].topo_version)
            stale.push_back(chains[i]);
    if (stale.empty())
        return LVBGPU_OK;
    if (uploaded)
        *uploaded = true;
    std::vector<uint32_t> bytes(stale.size(), 0);
    auto build = [&](int32_t i) {
        ChainSlot &cs = ctx->parked[(size_t)stale[(size_t)i]];
        char *slot = (char *)ctx->h_topo.p + (size_t)stale[(size_t)i] * ctx->gen_table_stride;
        cs.gen_K = kmax;
        bytes[(size_t)i] = ctx->gen_idx_bytes == 2 ? fill_tables<uint16_t>(cs.topo, (uint16_t *)slot, ctx->gen_table_stride / 2, kmax)
                                                   : fill_tables<int32_t>(cs.topo, (int32_t *)slot, ctx->gen_table_stride / 4, kmax);
    };
    int T = 1;
    if (stale.size() > 1)
    {
        if (!ctx->pool && host_threads() > 1)
            ctx->pool = new (std::nothrow) Pool(host_threads());
        if (ctx->pool)
            T = std::min<int>(ctx->pool->size(), (int)stale.size());
    }
    if (T == 1)
        for (size_t i = 0; i < stale.size(); i++)
            build((int32_t)i);
    else
        ctx->pool->run(T, [&](int t) {
            for (size_t i = (size_t)t; i < stale.size(); i += (size_t)T)
                build((int32_t)i);
        });
    // (a staging slot is only current for chains the HOST built last: accepted device moves rebuild their chain's
    // tables on the device, lvbgpu_chains_commit - so every stale chain is sent on its own, nothing in between)
    for (size_t i = 0; i < stale.size(); i++)
    {
        if (bytes[i] == 0)
            return ctx->fail(LVBGPU_E_ARG, "generator tables exceed their slot");
        ChainSlot &cs = ctx->parked[(size_t)stale[i]];
        cs.gen_table_bytes = bytes[i];
        cs.d_topo_version = cs.topo_version;
        const size_t off = (size_t)stale[i] * ctx->gen_table_stride;
        HIPCHK(ctx, hipMemcpyAsync((char *)ctx->d_topo4.p + off, (const char *)ctx->h_topo.p + off, bytes[i], hipMemcpyHostToDevice,
                                   ctx->stream));
    }
    return LVBGPU_OK;
}
} // namespace lvbgpu_detail

namespace lvbgpu_detail
{
// why a move named by the host cannot be made on this topology (nullptr: it can).  Same conditions
// as the generators (mutate_nni / mutate_spr / mutate_tbr, TreeOperations.c:174, 256-271, 450-461).
const char *move_defect(const Topology &t, const lvbgpu_move &m)
{
    const int32_t n = t.n, nb = t.nb, root = t.root;
    if (m.kind == 0)
        return (m.a >= n && m.a < nb) ? nullptr : "NNI needs an internal node";
    if (m.kind != 1 && m.kind != 2)
        return "kind must be 0 (NNI), 1 (SPR) or 2 (TBR)";
    const int32_t src = m.a, dest = m.b;
    if (src < 0 || src >= nb || dest < 0 || dest >= nb)
        return "node out of range";
    if (src == root || src == t.left[root] || src == t.right[root])
        return "the root and its children cannot be pruned";
    const int32_t sp = t.parent[src];
    const int32_t ss = t.left[sp] == src ? t.right[sp] : t.left[sp];
    if (dest == src || dest == sp || dest == ss || dest == root)
        return "destination is the source, its parent, its sister or the root";
    for (int32_t p = t.parent[dest]; p != UNSET; p = t.parent[p])
        if (p == src)
            return "destination lies inside the pruned subtree";
    if (m.kind == 2 && m.c >= 0)
    {
        const int32_t x = m.c;
        if (x >= n || x == t.left[src] || x == t.right[src])
            return "TBR re-roots at a leaf that is not a child of the subtree's top";
        bool inside = false;
        for (int32_t p = t.parent[x]; p != UNSET; p = t.parent[p])
            if (p == src)
            {
                inside = true;
                break;
            }
        if (!inside)
            return "TBR leaf lies outside the pruned subtree";
    }
    return nullptr;
}

// one generator launch + one walk over the candidates of k chains (draws[i].chain distinct, all with a resident
// tree); lengths_out holds the segments one after the other.  `moves` (host-named moves): k == 1 only.
// Two halves: submit (everything up to the lengths' read-back is enqueued; returns at once) and collect (waits for
// that batch alone and hands the lengths over).  Two batches may be in flight, in slots 0 and 1.
int propose_submit(lvbgpu_ctx *ctx, int32_t slot, int32_t k, const lvbgpu_chain_draw *draws, const lvbgpu_move *moves,
                   const lvbgpu_chain_rule *rules = nullptr)
{
    if (!ctx || slot < 0 || slot >= lvbgpu_ctx::PROP_SLOTS || k < 1 || k > (int32_t)MAX_GEN_SEGS || !draws || (moves && k != 1) ||
        (moves && rules))
        return LVBGPU_E_ARG;
    lvbgpu_ctx::PropSlot &ps = ctx->pslot[slot];
    if (ps.in_flight)
        return ctx->fail(LVBGPU_E_STATE, "that slot's batch has not been collected yet");
    if (ctx->pslot[1 - slot].in_flight && (rules || ctx->pslot[1 - slot].step))
        return ctx->fail(LVBGPU_E_STATE, "a step and another batch may not be in flight together");
    if (ctx->n < 5)
        return ctx->fail(LVBGPU_E_ARG, "rearrangements need at least 5 taxa");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    AllParked guard(ctx);
    if (ctx->follow.pending)
    {
        // the generator needs versions, roots (no move changes a root) and device tables - not the host's topologies,
        // unless a chain's tables have to be built from one here, or the moves are named against one
        bool need = moves != nullptr;
        for (int32_t i = 0; i < k && !need; i++)
            if (draws[i].chain >= 0 && draws[i].chain < ctx->nchains)
                need = ctx->parked[(size_t)draws[i].chain].d_topo_version != ctx->parked[(size_t)draws[i].chain].topo_version;
        if (need)
        {
            const int rf = resolve_follow(ctx);
            if (rf != LVBGPU_OK)
                return rf;
        }
    }
    int64_t total = 0;
    std::vector<int32_t> chains((size_t)k);
    uint64_t seen = 0;
    for (int32_t i = 0; i < k; i++)
    {
        const lvbgpu_chain_draw &d = draws[i];
        if (d.chain < 0 || d.chain >= ctx->nchains || d.count < 1 || d.kind < -3 || d.kind > 2 || ((seen >> d.chain) & 1u))
            return ctx->fail(LVBGPU_E_ARG, "draw " + std::to_string(i) + ": chain out of range or listed twice, or bad count / kind");
        if (!ctx->parked[(size_t)d.chain].have_tree)
            return ctx->fail(LVBGPU_E_STATE, "no resident tree: call lvbgpu_set_tree first");
        seen |= 1ull << d.chain;
        chains[(size_t)i] = d.chain;
        total += d.count;
    }
    // fixed strides: a program has at most (n-3)+3 tokens; edits are capped (longer TBR paths overflow)
    const uint32_t stride_t = (uint32_t)ctx->n + 8u;
    const uint32_t stride_e = (uint32_t)std::min<int64_t>(ctx->nb, 512);
    if ((uint64_t)total * stride_t >= (1ull << 32) || (uint64_t)total * ctx->ntiles >= (1ull << 31))
        return ctx->fail(LVBGPU_E_ARG, "batch too large");
    const int32_t B = (int32_t)total;
    int rc = prepare_tables(ctx, chains.data(), k);
    if (rc != LVBGPU_OK)
        return rc;
    if (!ps.batch)
    {
        ps.batch = new (std::nothrow) lvbgpu_batch();
        if (!ps.batch)
            return LVBGPU_E_NOMEM;
    }
    if (!ps.done_ev)
        HIPCHK(ctx, hipEventCreateWithFlags(&ps.done_ev, hipEventDisableTiming));
    if (!ps.walk_ev)
        HIPCHK(ctx, hipEventCreateWithFlags(&ps.walk_ev, hipEventDisableTiming));
    lvbgpu_batch *bt = ps.batch;
    const size_t o_t = align16((size_t)B * sizeof(CandDesc));
    const size_t o_d = o_t + align16((size_t)B * stride_t * 4);
    const size_t bytes = o_d + align16((size_t)B * stride_t * 4);
    HIPCHK(ctx, bt->d_prog.reserve(bytes));
    HIPCHK(ctx, bt->d_len.reserve((size_t)B * 8));
    HIPCHK(ctx, bt->h_len.reserve((size_t)B * 8));
    HIPCHK(ctx, ps.d_pedits.reserve((size_t)B * stride_e * sizeof(lvbgpu_edit_dev)));
    HIPCHK(ctx, ps.d_pinfo.reserve((size_t)B * sizeof(ProposalInfo)));
    bt->ctx = ctx;
    bt->B = B;
    bt->off_toks = o_t;
    bt->off_dsts = o_d;
    bt->full_mode = false;
    bt->spans_chains = true; // its programs name their own chains; each segment's tree version is kept in p_segs
    bt->stats = lvbgpu_batch_stats{};
    bt->stats.candidates = B;
    bt->stats.max_stack = 1; // at most one sibling set waits while the other path is walked
    ctx->p_stride_t = stride_t;
    ctx->p_stride_e = stride_e;
    ps.p_B = 0;
    ps.B = B;
    ps.segs.clear();
    const lvbgpu_move_dev *d_moves = nullptr;
    if (moves)
    {
        static_assert(sizeof(lvbgpu_move) == sizeof(lvbgpu_move_dev), "move layout");
        const Topology &topo = ctx->parked[(size_t)draws[0].chain].topo;
        // admissibility is an O(depth) walk per move: spread it over the host threads for long batches
        int T = 1;
        if (B >= 8192) // measured: waking the pool costs more than it saves below that
        {
            if (!ctx->pool && host_threads() > 1)
                ctx->pool = new (std::nothrow) Pool(host_threads());
            if (ctx->pool)
                T = std::max(1, std::min(ctx->pool->size(), B / 1024));
        }
        std::vector<int32_t> first_bad((size_t)T, -1);
        auto check = [&](int t) {
            for (int32_t b = (int32_t)((int64_t)B * t / T); b < (int32_t)((int64_t)B * (t + 1) / T); b++)
                if (move_defect(topo, moves[b]))
                {
                    first_bad[(size_t)t] = b;
                    return;
                }
        };
        if (T == 1)
            check(0);
        else
            ctx->pool->run(T, check);
        for (int t = 0; t < T; t++)
            if (first_bad[(size_t)t] >= 0)
            {
                const int32_t b = first_bad[(size_t)t];
                return ctx->fail(LVBGPU_E_TOPOLOGY, "move " + std::to_string(b) + ": " + move_defect(topo, moves[b]));
            }
        HIPCHK(ctx, ctx->d_moves.reserve((size_t)B * sizeof(lvbgpu_move)));
        HIPCHK(ctx, ctx->h_moves.reserve((size_t)B * sizeof(lvbgpu_move)));
        memcpy(ctx->h_moves.p, moves, (size_t)B * sizeof(lvbgpu_move)); // pinned staging: the caller's array may go away
        // a short list is read by the generator where it lies (16 bytes per wave, once); a long one is copied
        if (ctx->direct_steps && (size_t)B * sizeof(lvbgpu_move) <= DIRECT_READ_MAX_BYTES)
            d_moves = (const lvbgpu_move_dev *)ctx->h_moves.p;
        else
        {
            HIPCHK(ctx, hipMemcpyAsync(ctx->d_moves.p, ctx->h_moves.p, (size_t)B * sizeof(lvbgpu_move),
                                       hipMemcpyHostToDevice, ctx->stream));
            d_moves = (const lvbgpu_move_dev *)ctx->d_moves.p;
        }
    }
    static const bool allow_watcher = [] {
        const char *e = getenv("LVBGPU_WATCHER");
        return !(e && e[0] == '0');
    }();
    // (pipelined batches of a small launch - below WATCH_PIPELINED_MAX_ITEMS waves - take the watcher too: the copy
    // stream's two events and its copy cost such a step more than the hand-over lengthens its walk: 64 x 10k NNI,
    // B = 1024: 34 -> 61 M candidates/s; 500 x 50k, B = 256: 8.2 -> 9.4, B = 1024: +4 %; at B = 4096 the two are equal
    // within 1 % and the walk is 1.6 us shorter without the hand-over)
    constexpr uint64_t WATCH_PIPELINED_MAX_ITEMS = 32768;
    const uint32_t groups_now = choose_groups((uint32_t)B, ctx->ntiles, ctx->target_waves);
    ps.watched = allow_watcher && groups_now <= WATCH_MAX_GROUPS &&
                 (!ctx->pslot[1 - slot].in_flight || (uint64_t)B * groups_now < WATCH_PIPELINED_MAX_ITEMS);
    // A STEP: the rule by which every chain accepts rides with the batch; the library decides at the collect
    // (lvb_amd/csrc/decide.h) and commits every chain's first taken candidate there.
    ps.step = rules != nullptr;
    ps.host_rules.clear();
    if (rules)
    {
        uint32_t st = 0;
        for (int32_t i = 0; i < k; i++)
        {
            ps.host_rules.push_back(DecideRule{(long long)rules[i].cur_length, rules[i].temperature, rules[i].min_len_tree,
                                               (unsigned long long)rules[i].accept_seed, st, (uint32_t)draws[i].count});
            st += (uint32_t)draws[i].count;
        }
    }
    GenArgs ga{};
    ga.tables = ctx->d_topo4.p;
    ga.idx_bytes = ctx->gen_idx_bytes;
    ga.table_stride = ctx->gen_table_stride;
    ga.K = ctx->gen_kmax;
    ga.n = (int32_t)ctx->n;
    ga.nb = ctx->nb;
    ga.leaf_order_len = (uint32_t)ctx->n;
    ga.stride_t = stride_t;
    ga.stride_e = stride_e;
    ga.toks = (uint32_t *)((char *)bt->d_prog.p + o_t);
    ga.dsts = (int32_t *)((char *)bt->d_prog.p + o_d);
    ga.edits = (lvbgpu_edit_dev *)ps.d_pedits.p;
    ga.cands = (CandDesc *)bt->d_prog.p;
    ga.info = (ProposalInfo *)ps.d_pinfo.p;
    ga.len_out = (unsigned long long *)bt->d_len.p;
    // two candidates per wave (fitch_walk_pair, LVBGPU_PAIR=n: every batch of n candidates and more): a wave that walks two
    // programs loads the rows of their common end once.  Who walks with whom is settled by the generating workgroups among
    // the sixteen candidates each has just drawn (GenArgs::pairs): nothing waits, nothing is sorted, and the generator's
    // launch takes 4 us longer (9.9 -> 14.5 us at B = 4096: the workgroup's slowest candidate, then the pairing).  What that
    // shares depends on the moves and the tree: NNI neighbours 25-33 % of their row reads, SPR / TBR neighbours 10 %.
    // Measured, 500 x 50 000, B = 4096, step without / with: SPR on a fresh tree (D = 20) 102.0 / 106.8 us; NNI on a fresh
    // tree (D = 11) 72.2 / 76.0; NNI on a tree mixed by 3000 moves (D = 36) 144.5 / 126.0 (walk 133.6 -> 111.2); the 32-chain
    // annealing run 1.24-1.27 / 1.20-1.23 s.  Off unless asked for: it pays where paths are long AND run together, which the
    // library cannot tell from what it is handed.
    const bool pair_up = ctx->pair_min > 0 && B >= ctx->pair_min;
    bt->npairs = 0;
    if (pair_up)
    {
        uint32_t np = 0;
        for (int32_t i = 0; i < k; i++)
            np += ((uint32_t)draws[i].count + 1u) / 2u;
        HIPCHK(ctx, bt->d_pairs.reserve((size_t)np * 8));
        ga.pairs = (uint32_t *)bt->d_pairs.p;
        bt->npairs = np;
    }
    ga.moves = d_moves;
    static const bool gen_profile = getenv("LVBGPU_GEN_PROFILE") != nullptr;
    if (gen_profile)
    {
        HIPCHK(ctx, ctx->d_gen_prof.reserve(256 * 8 * 8));
        ga.prof = (unsigned long long *)ctx->d_gen_prof.p;
    }
    ga.nseg = (uint32_t)k;
    uint32_t start = 0;
    for (int32_t i = 0; i < k; i++)
    {
        const lvbgpu_chain_draw &d = draws[i];
        const ChainSlot &cs = ctx->parked[(size_t)d.chain];
        GenSeg &sg = ga.seg[i];
        sg.start = start;
        sg.count = (uint32_t)d.count;
        sg.kind_all = (int8_t)d.kind;
        sg.mix_a = d.mix_a;
        sg.mix_b = d.mix_b;
        sg.seed_lo = (uint32_t)d.seed;
        sg.seed_hi = (uint32_t)(d.seed >> 32);
        sg.root = cs.topo.root;
        sg.chain = (uint8_t)d.chain;
        ga.table_bytes = std::max(ga.table_bytes, cs.gen_table_bytes); // (the same for every tree of these taxa)
        ps.segs.push_back({d.chain, (int32_t)start, d.count, cs.topo_version});
        start += (uint32_t)d.count;
    }
    // (While the other slot's batch is on the device this batch's generator COULD run beside that batch's walk, on a
    // stream of its own: measured at B = 4096 the walk then takes 100 us instead of 88 - the generator's workgroups
    // hold LDS and wave slots - and the step gains nothing over queueing behind it; with that stream at the lowest
    // priority a step takes 161 us.  profiles/experiments/r02_walk_and_step.md)
    // What the chains' last commits and re-roots left pending goes out NOW, and the generator with it: one post launch
    // (commit walk + table rebuilds + this generator, whose segments wait for their chains' rebuilds) instead of up to
    // five launches on two streams.  Not when the accepted candidates' programs lie in this very slot's buffers, which the
    // generator is about to overwrite, or when the moves are named by the host (their list is read beside the tables).
    if (ctx->pend.any())
    {
        const bool same_buffers = ctx->pend.k_pick > 0 && ctx->pend.src_slot == slot;
        if (!same_buffers && !moves && post_can_generate(ga))
            rc = flush_pending(ctx, &ga);
        else
        {
            rc = flush_pending(ctx, nullptr);
            if (rc == LVBGPU_OK)
                HIPCHK(ctx, launch_propose(ga, ctx->stream));
        }
        if (rc != LVBGPU_OK)
        {
            ps.segs.clear();
            return rc;
        }
    }
    else
        HIPCHK(ctx, launch_propose(ga, ctx->stream));
    bt->len_zeroed = true; // by the generator
    // Only the lengths come back per step (a move's descriptor and edits are fetched when, and only when, the caller
    // wants that candidate - lvbgpu_proposal_edits - or accepts it - lvbgpu_chains_commit).  One batch at a time the
    // walk hands them over itself: watcher waves at the end of its grid store them into the batch's pinned buffer and
    // release this slot's flag (WalkArgs::watcher) - no copy behind the walk for the step to wait for (B = 4096: 135 ->
    // 121 us per step).  With the other slot's batch in flight the copy is hidden anyway (it goes to the copy stream,
    // beside the next batch's generator: in the main stream it held that generator back for 21 us, 34 -> 38.6 M
    // candidates/s), and the walk stays 1.6 us shorter without its hand-over.  LVBGPU_WATCHER=0: always the copy (A/B
    // runs); launches of more than WATCH_MAX_GROUPS tile groups keep it too.
    if (ps.watched)
    {
        if (!ps.h_flag.p)
        {
            HIPCHK(ctx, ps.h_flag.reserve(WATCH_WAVES * 4)); // one word per watcher wave
            memset(ps.h_flag.p, 0, WATCH_WAVES * 4);
        }
        bt->watch_flag = (uint32_t *)ps.h_flag.p;
        bt->watch_seq = ++ps.seq == 0xFFFFFFFFu ? (ps.seq = 1) : ps.seq; // 0xFFFFFFFF is the watcher's "gave up"
    }
    else
        bt->watch_flag = nullptr;
    rc = lvbgpu_batch_launch(ctx, bt);
    if (rc != LVBGPU_OK)
    {
        ps.segs.clear();
        return rc;
    }
    if (!ps.watched)
    {
        // the copy: beside the next batch's generator when the other slot's batch is in flight, else in the main stream
        // (the hop to another stream costs a lone step 13 us)
        if (ctx->pslot[1 - slot].in_flight)
        {
            HIPCHK(ctx, hipEventRecord(ps.walk_ev, ctx->stream));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_stream, ps.walk_ev, 0));
            HIPCHK(ctx, hipMemcpyAsync(bt->h_len.p, bt->d_len.p, (size_t)B * 8, hipMemcpyDeviceToHost, ctx->copy_stream));
            HIPCHK(ctx, hipEventRecord(ps.done_ev, ctx->copy_stream));
        }
        else
        {
            HIPCHK(ctx, hipMemcpyAsync(bt->h_len.p, bt->d_len.p, (size_t)B * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipEventRecord(ps.done_ev, ctx->stream));
        }
    }
    ps.in_flight = true;
    ps.submit_ord = ++ctx->submits;
    if (k == 1 && draws[0].chain == guard.sel)
        ps.p_B = B; // lvbgpu_proposal_edits may name its candidates (slot 0, once collected)
    return LVBGPU_OK;
}

int propose_collect(lvbgpu_ctx *ctx, int32_t slot, int64_t *lengths_out)
{
    if (!ctx || slot < 0 || slot >= lvbgpu_ctx::PROP_SLOTS || !lengths_out)
        return LVBGPU_E_ARG;
    lvbgpu_ctx::PropSlot &ps = ctx->pslot[slot];
    if (!ps.in_flight)
        return ctx->fail(LVBGPU_E_STATE, "nothing was submitted in that slot");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ps.in_flight = false;
    // this batch alone: whatever was enqueued behind it (the other slot's batch, commits) is not waited for
    const WaitClock clock(ctx->wait_limit_s);
    if (ps.watched)
    {
        const uint32_t *flag = (const uint32_t *)ps.h_flag.p;
        // every watcher wave sets its own word: all of them, in any order (a word still behind is re-read, the ones
        // before it were seen complete)
        // (Nothing but memory is polled here.  The loop used to ask hipStreamQuery every few thousand spins whether the
        // stream had drained or failed; measured on MI355X / ROCm 7.2 that call can BLOCK until the kernel at the head
        // of the stream has finished - with a 400 ms kernel in front it returned after 400 ms - and it serialises
        // host threads that drive other contexts.  A wait that cannot be trusted to return cannot be bounded by it: the
        // clock alone ends this one.)
        uint32_t seen = ps.seq, w = 0;
        for (uint32_t spins = 1; w < WATCH_WAVES; spins++)
        {
            seen = __atomic_load_n(flag + w, __ATOMIC_ACQUIRE);
            if (seen == ps.seq)
            {
                w++;
                continue;
            }
            if (seen == 0xFFFFFFFFu)
                break;
            if ((spins & 4095u) == 0 && clock.expired())
            {
                ps.segs.clear();
                ps.p_B = 0;
                return ctx->fail_wait("lvbgpu_chains_collect: the walk's watcher waves", clock.waited());
            }
        }
        if (seen != ps.seq)
        {
            ps.segs.clear();
            ps.p_B = 0;
            return ctx->fail(LVBGPU_E_STATE, "the walk's watcher did not hand the lengths over (flag " + std::to_string(seen) + ")");
        }
    }
    else
    {
        // small batches are polled for (the runtime's wake-up costs ~10 us), big ones sleep between polls; both give up
        // at the context's wait limit
        hipError_t q;
        uint32_t spins = 0;
        while ((q = hipEventQuery(ps.done_ev)) == hipErrorNotReady)
        {
            if (ps.B > SPIN_WAIT_MAX_B)
                std::this_thread::sleep_for(std::chrono::microseconds(20));
            if ((++spins & 1023u) == 0 && clock.expired())
            {
                ps.segs.clear();
                ps.p_B = 0;
                return ctx->fail_wait("lvbgpu_chains_collect: the lengths' read-back", clock.waited());
            }
        }
        HIPCHK(ctx, q);
    }
    ctx->last_slot = slot;
    ctx->collected_ord = std::max(ctx->collected_ord, ps.submit_ord); // (take_pick_slot)
    const int32_t B = ps.B;
    const int64_t *len = (const int64_t *)ps.batch->h_len.p;
    for (int32_t b = 0; b < B; b++)
    {
        if (len[b] >= PROPOSAL_OVERFLOW_LENGTH)
        {
            lengths_out[b] = INT64_MAX;
            continue;
        }
        lengths_out[b] = len[b];
        if (len[b] <= 0)
        {
            ps.segs.clear();
            ps.p_B = 0;
            return ctx->fail(LVBGPU_E_ZEROLEN, "assertion failed: changes > 0 (device-built candidate " + std::to_string(b) +
                                                   " of " + std::to_string(B) + " scored " + std::to_string(len[b]) + ")");
        }
    }
    return LVBGPU_OK;
}

int propose_core(lvbgpu_ctx *ctx, int32_t k, const lvbgpu_chain_draw *draws, int64_t *lengths_out, const lvbgpu_move *moves)
{
    if (!lengths_out)
        return LVBGPU_E_ARG;
    if (ctx && ctx->pslot[0].in_flight)
        return ctx->fail(LVBGPU_E_STATE, "slot 0 holds a submitted batch: collect it first");
    int rc = propose_submit(ctx, 0, k, draws, moves);
    if (rc == LVBGPU_OK)
        rc = propose_collect(ctx, 0, lengths_out);
    return rc;
}

// the single-tree forms: the selected chain, one segment
int propose_score_impl(lvbgpu_ctx *ctx, int32_t B, int32_t kind, uint32_t mix_a, uint32_t mix_b, uint64_t seed,
                       int64_t *lengths_out, const lvbgpu_move *moves = nullptr)
{
    if (!ctx || B < 1 || kind < -3 || kind > 2 || !lengths_out)
        return LVBGPU_E_ARG;
    if (!ctx->have_tree)
        return ctx->fail(LVBGPU_E_STATE, "no resident tree: call lvbgpu_set_tree first");
    const lvbgpu_chain_draw d{ctx->chain, B, kind, mix_a, mix_b, seed};
    return propose_core(ctx, 1, &d, lengths_out, moves);
}
} // namespace lvbgpu_detail

extern "C" int lvbgpu_propose_score(lvbgpu_ctx *ctx, int32_t B, int32_t kind, uint64_t seed, int64_t *lengths_out)
{
    if (kind < -1)
        return LVBGPU_E_ARG;
    return propose_score_impl(ctx, B, kind, 0, 0, seed, lengths_out);
}

extern "C" int lvbgpu_score_moves(lvbgpu_ctx *ctx, int32_t B, const lvbgpu_move *moves, int64_t *lengths_out)
{
    if (!moves)
        return LVBGPU_E_ARG;
    return propose_score_impl(ctx, B, 0, 0, 0, 0, lengths_out, moves);
}

extern "C" int lvbgpu_propose_score_mixed(lvbgpu_ctx *ctx, int32_t B, double p_nni, double p_spr, int64_t parity,
                                          uint64_t seed, int64_t *lengths_out)
{
    if (parity >= 0)
        return propose_score_impl(ctx, B, -2, (uint32_t)(parity & 1), 0, seed, lengths_out);
    if (!(p_nni >= 0) || !(p_spr >= 0) || p_nni + p_spr > 1.0 + 1e-12)
        return LVBGPU_E_ARG;
    auto scaled = [](double p) { return (uint32_t)std::min(4294967295.0, p * 4294967296.0); };
    return propose_score_impl(ctx, B, -3, scaled(p_nni), scaled(p_nni + p_spr), seed, lengths_out);
}

extern "C" int lvbgpu_chains_propose_score(lvbgpu_ctx *ctx, int32_t k, const lvbgpu_chain_draw *draws, int64_t *lengths_out)
{
    return propose_core(ctx, k, draws, lengths_out, nullptr);
}

// diagnostic (LVBGPU_POST_PROFILE set): the clock stamps of the last post launch's workgroups: out[0] = their number, then
// {role (1 rebuild, 2 commit walk, 3 generator), start, -, end} per workgroup (100 MHz clock), the first 1000
extern "C" int lvbgpu_debug_post_stamps(lvbgpu_ctx *ctx, unsigned long long *out4065)
{
    if (!ctx || !out4065 || !ctx->d_post_prof.p)
        return LVBGPU_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(out4065, ctx->d_post_prof.p, (1 + 4 * 1000 + 64) * 8, hipMemcpyDeviceToHost));
    return LVBGPU_OK;
}

// diagnostic: who walked with whom in the last device-built batch of `slot` (two candidates per wave): out[2 p], out[2 p + 1]
// = the candidates of pair p (0xFFFFFFFF: walked alone); *npairs = 0 when that batch was walked one candidate per wave
extern "C" int lvbgpu_debug_pairs(lvbgpu_ctx *ctx, int32_t slot, uint32_t *out, int32_t cap_pairs, int32_t *npairs)
{
    if (!ctx || slot < 0 || slot >= lvbgpu_ctx::PROP_SLOTS || !out || !npairs || !ctx->pslot[slot].batch)
        return LVBGPU_E_ARG;
    const lvbgpu_batch *bt = ctx->pslot[slot].batch;
    *npairs = (int32_t)bt->npairs;
    if (bt->npairs == 0)
        return LVBGPU_OK;
    if ((uint32_t)cap_pairs < bt->npairs)
        return LVBGPU_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(out, bt->d_pairs.p, (size_t)bt->npairs * 8, hipMemcpyDeviceToHost));
    return LVBGPU_OK;
}

// diagnostic (LVBGPU_GEN_PROFILE set): the clock stamps the last generator launch left for its first 256 candidates
extern "C" int lvbgpu_debug_generator_stamps(lvbgpu_ctx *ctx, unsigned long long *out2048)
{
    if (!ctx || !out2048 || !ctx->d_gen_prof.p)
        return LVBGPU_E_ARG;
    ENTER(ctx);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(out2048, ctx->d_gen_prof.p, 256 * 8 * 8, hipMemcpyDeviceToHost));
    return LVBGPU_OK;
}

extern "C" int lvbgpu_chains_submit(lvbgpu_ctx *ctx, int32_t slot, int32_t k, const lvbgpu_chain_draw *draws)
{
    return propose_submit(ctx, slot, k, draws, nullptr);
}

extern "C" int lvbgpu_chains_ready(lvbgpu_ctx *ctx, int32_t slot, int32_t *ready)
{
    if (!ctx || slot < 0 || slot >= lvbgpu_ctx::PROP_SLOTS || !ready)
        return LVBGPU_E_ARG;
    lvbgpu_ctx::PropSlot &ps = ctx->pslot[slot];
    if (!ps.in_flight)
        return ctx->fail(LVBGPU_E_STATE, "nothing was submitted in that slot");
    *ready = 0;
    if (ps.watched)
    {
        // (memory only, as the collect: every watcher wave's word, a "gave up" word counts as done - the collect reports it)
        const uint32_t *flag = (const uint32_t *)ps.h_flag.p;
        for (uint32_t w = 0; w < WATCH_WAVES; w++)
        {
            const uint32_t seen = __atomic_load_n(flag + w, __ATOMIC_ACQUIRE);
            if (seen == 0xFFFFFFFFu)
                break;
            if (seen != ps.seq)
                return LVBGPU_OK;
        }
        *ready = 1;
        return LVBGPU_OK;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const hipError_t q = hipEventQuery(ps.done_ev);
    if (q != hipSuccess && q != hipErrorNotReady)
        return ctx->fail_hip(q, "hipEventQuery(done_ev)");
    *ready = q == hipSuccess ? 1 : 0;
    return LVBGPU_OK;
}

extern "C" int lvbgpu_chains_collect(lvbgpu_ctx *ctx, int32_t slot, int64_t *lengths_out)
{
    if (ctx && slot >= 0 && slot < lvbgpu_ctx::PROP_SLOTS && ctx->pslot[slot].in_flight && ctx->pslot[slot].step)
        return ctx->fail(LVBGPU_E_STATE, "that slot holds a step: collect it with lvbgpu_chains_step_collect");
    return propose_collect(ctx, slot, lengths_out);
}

namespace lvbgpu_detail
{
// The next pinned slot for a commit's or a re-root's picks / programs.  Its readers - the commit walk (main stream),
// the table rebuild and the gather (side stream) - are all finished once a batch SUBMITTED AFTER this use has been
// collected: that batch's walk was enqueued behind the commit walk, and its generator waited for the rebuild.  (A batch
// submitted BEFORE the use proves nothing: it may have been collected while the commit walk was still queued behind
// it.)  So slots need no events of their own (an event record + a wait cost the host 4-5 us per commit, where the host
// is what the device waits for): every use remembers how many batches had been submitted before it, and a slot whose
// use no later batch has come back for drains the streams before it is written again.
int resolve_follow(lvbgpu_ctx *ctx)
{
    lvbgpu_ctx::Follow &f = ctx->follow;
    if (!f.pending)
        return LVBGPU_OK;
    if (ctx->pend.k_pick) // the records are sent by the post launch, which has not been given to the device yet
    {
        const int rp = flush_pending(ctx, nullptr);
        if (rp != LVBGPU_OK)
            return rp;
    }
    f.pending = false;
    const uint32_t out_stride = (uint32_t)align16(sizeof(ProposalInfo) + (size_t)ctx->p_stride_e * sizeof(lvbgpu_edit_dev));
    const size_t o_out = 64 + align16((size_t)MAX_CHAINS * 4);
    const char *h = (const char *)ctx->h_pick[f.slot].p;
    const uint32_t *flag = (const uint32_t *)h;
    const WaitClock clock(ctx->wait_limit_s);
    for (uint32_t spins = 1; __atomic_load_n(flag, __ATOMIC_ACQUIRE) != f.seq; spins++) // (memory only, see propose_collect)
        if ((spins & 1023u) == 0 && clock.expired())
        {
            for (int32_t j = 0; j < f.k; j++) // the device may or may not have walked them: not trustworthy any more
                if (f.has[j])
                    ctx->parked[(size_t)f.chains[j]].have_tree = false;
            return ctx->fail_wait("lvbgpu_chains_commit: the picked moves' gather", clock.waited());
        }
    // what lvbgpu_chains_picked_edits hands out is a COPY: the pinned slot may be taken again (by the next commits and
    // re-roots) long before the caller asks for the moves
    ctx->pick_records.assign(h + o_out, h + o_out + (size_t)f.k * out_stride);
    ctx->pick_records_stride = out_stride;
    for (int32_t j = 0; j < f.k; j++)
    {
        if (!f.has[j])
            continue;
        const char *rec = h + o_out + (size_t)j * out_stride;
        const ProposalInfo pi = *(const ProposalInfo *)rec;
        ChainSlot &cs = ctx->parked[(size_t)f.chains[j]];
        std::string why;
        if (pi.overflow || !ctx->pb.apply_edits(cs.topo, (const Edit *)(rec + sizeof(ProposalInfo)), pi.n_edits, -1, &why))
        {
            cs.have_tree = false; // the device has walked it: this chain's resident state is no longer trustworthy
            return ctx->fail(LVBGPU_E_TOPOLOGY, "chain " + std::to_string(f.chains[j]) + ": " + (pi.overflow ? "overflowed candidate" : why));
        }
    }
    return LVBGPU_OK;
}

int settle(lvbgpu_ctx *ctx)
{
    if (!ctx->follow.pending)
        return LVBGPU_OK;
    AllParked guard(ctx);
    return resolve_follow(ctx);
}

hipError_t take_pick_slot(lvbgpu_ctx *ctx, int *slot)
{
    // (not the slot whose records the host has yet to read: lvbgpu_ctx::follow)
    if (ctx->follow.pending && ctx->follow.slot == ctx->pick_slot)
        ctx->pick_slot = (ctx->pick_slot + 1) % lvbgpu_ctx::PICK_SLOTS;
    const int s = ctx->pick_slot;
    if (ctx->pick_used[s] && ctx->collected_ord <= ctx->pick_use_ord[s])
    {
        const hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess)
            return e;
        for (int i = 0; i < lvbgpu_ctx::PICK_SLOTS; i++)
            ctx->pick_used[i] = false; // everything enqueued so far is done
    }
    ctx->pick_used[s] = true;
    ctx->pick_use_ord[s] = ctx->submits;
    *slot = s;
    ctx->pick_slot = (s + 1) % lvbgpu_ctx::PICK_SLOTS;
    return hipSuccess;
}

// (A table rebuild rewrites the generator's tables of its chains in place; it runs in the post launch on the MAIN stream,
// behind every generator that was submitted before it - a batch in the other slot may have been drawn from the same
// chain - and in front of every later one: no generator ever reads torn tables.)
// lvbgpu_proposal_edits names candidates of slot 0's last single-chain batch relative to the tree they were drawn
// from: once that chain's tree has moved (and its device tables with it, so the version test there no longer sees it)
// their rewrites mean nothing
void forget_named_candidates(lvbgpu_ctx *ctx, uint64_t chain_mask)
{
    lvbgpu_ctx::PropSlot &p0 = ctx->pslot[0];
    if (p0.p_B > 0 && p0.segs.size() == 1 && ((chain_mask >> p0.segs[0].chain) & 1u))
        p0.p_B = 0;
}

} // namespace lvbgpu_detail


namespace lvbgpu_detail
{
// the layout of a pinned slot that receives picked candidates' records: [flag][gap][k x (descriptor + rewrites)]
static inline uint32_t pick_record_stride(const lvbgpu_ctx *ctx)
{
    return (uint32_t)align16(sizeof(ProposalInfo) + (size_t)ctx->p_stride_e * sizeof(lvbgpu_edit_dev));
}
static constexpr size_t PICK_RECORDS_AT = 64 + ((size_t)MAX_CHAINS * 4 + 15) / 16 * 16;

// Give the device what lvbgpu_ctx::pend holds, as ONE post launch on the main stream (kernels.hpp PostArgs): the commit
// walk of the accepted candidates' own device-built programs and of the host-built programs (re-roots, host-made
// candidates), the table rebuilds of the chains that moved (wave 0 of an accepted candidate's workgroup first sends its
// record to the host), and - gen != null - the NEXT batch's generator, whose segments wait for their chains' rebuilds.
int flush_pending(lvbgpu_ctx *ctx, const GenArgs *gen)
{
    lvbgpu_ctx::Pending &pd = ctx->pend;
    if (!pd.any() && !gen)
        return LVBGPU_OK;
    const lvbgpu_ctx::Pending q = pd; // (whatever happens below, nothing stays pending)
    pd = lvbgpu_ctx::Pending{};
    PostArgs pa{};
    const int32_t k = q.k_pick + q.k_ext;
    const char *hx = q.k_ext ? (const char *)ctx->h_pick[q.ext_slot].p : nullptr;
    if (k)
    {
        const size_t old_done = ctx->d_done.cap;
        HIPCHK(ctx, ctx->d_done.reserve((size_t)(MAX_CHAINS + 1) * 4));
        if (ctx->d_done.cap != old_done) // once per context
        {
            HIPCHK(ctx, hipMemsetAsync(ctx->d_done.p, 0, ctx->d_done.cap, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        }
        HIPCHK(ctx, ctx->d_tmp_changes.reserve((size_t)MAX_CHAINS * (size_t)(ctx->nb + 1) * 8));
        if (ctx->tmp_changes_zeroed_cap != ctx->d_tmp_changes.cap)
        {
            HIPCHK(ctx, hipMemsetAsync(ctx->d_tmp_changes.p, 0, ctx->d_tmp_changes.cap, ctx->stream));
            ctx->tmp_changes_zeroed_cap = ctx->d_tmp_changes.cap;
        }
        HIPCHK(ctx, ctx->d_len.reserve(8));
        // the host-built programs: read where they lie (pinned) while that is little, else copied first
        const void *xprog = hx;
        if (q.k_ext)
        {
            const uint32_t ngroups = choose_groups((uint32_t)k, ctx->ntiles, ctx->target_waves);
            if (!(ctx->direct_steps && q.ext_o_x * ngroups <= DIRECT_READ_MAX_BYTES))
            {
                HIPCHK(ctx, ctx->d_commit[0].reserve(q.ext_o_x));
                HIPCHK(ctx, hipMemcpyAsync(ctx->d_commit[0].p, hx, q.ext_o_x, hipMemcpyHostToDevice, ctx->stream));
                xprog = ctx->d_commit[0].p;
            }
        }
        WalkArgs a;
        if (q.k_pick)
        {
            lvbgpu_batch *bt = ctx->pslot[q.src_slot].batch;
            a = resident_args(ctx, bt->d_prog.p, bt->off_toks, bt->off_dsts, ctx->d_len.p, (uint32_t)k, std::max(1, q.ext_max_stack));
            a.use_pick = 1;
            memcpy(a.pick_idx, q.where, (size_t)q.k_pick * 4);
            if (q.k_ext)
            {
                a.n_first = (uint32_t)q.k_pick;
                a.cands2 = (const CandDesc *)xprog;
                a.toks2 = (const uint32_t *)((const char *)xprog + q.ext_o_t);
                a.dsts2 = (const int32_t *)((const char *)xprog + q.ext_o_d);
            }
        }
        else
            a = resident_args(ctx, xprog, q.ext_o_t, q.ext_o_d, ctx->d_len.p, (uint32_t)k, q.ext_max_stack);
        a.s_all_out = (unsigned long long *)ctx->d_scalars;
        a.tmp_changes = (unsigned long long *)ctx->d_tmp_changes.p;
        a.tmp_stride = (uint32_t)(ctx->nb + 1);
        a.done_count = (uint32_t *)ctx->d_done.p;
        pa.commit = a;
    }
    // the tables: accepted candidates' chains (rewrites from the batch they were drawn in), then the host-named moves'
    const bool reb_picks = q.k_pick && q.pick_rebuild, reb_ext = q.k_ext && q.ext_rebuild;
    uint64_t rebuilt = 0;
    if (q.k_pick || reb_ext)
    {
        RebuildArgs &ra = pa.reb;
        ra.tables = ctx->d_topo4.p;
        ra.table_stride = ctx->gen_table_stride;
        ra.idx_bytes = ctx->gen_idx_bytes;
        ra.n = (int32_t)ctx->n;
        ra.nb = ctx->nb;
        ra.K = ctx->gen_kmax;
        ra.leaf_order_len = (uint32_t)ctx->n;
        ra.table_bytes = ctx->parked.empty() ? 0u : ctx->parked[0].gen_table_bytes; // (the same for every tree of these taxa)
        for (const ChainSlot &cs : ctx->parked)
            ra.table_bytes = std::max(ra.table_bytes, cs.gen_table_bytes);
        ra.n_pick = (uint32_t)q.k_pick;
        ra.rebuild_picks = reb_picks ? 1u : 0u;
        pa.n_reb = (uint32_t)q.k_pick + (reb_ext ? (uint32_t)q.k_ext : 0u);
        if (q.k_pick)
        {
            lvbgpu_ctx::PropSlot &ps = ctx->pslot[q.src_slot];
            memcpy(ra.pick_idx, q.where, (size_t)q.k_pick * 4);
            ra.cands = (const CandDesc *)ps.batch->d_prog.p;
            ra.info = (const ProposalInfo *)ps.d_pinfo.p;
            ra.edits = (const lvbgpu_edit_dev *)ps.d_pedits.p;
            ra.stride_e = ctx->p_stride_e;
            // ... and their records on the way to the host (the host follows the moves while the device works)
            GatherArgs &gat = pa.gat;
            char *h = (char *)ctx->h_pick[q.gather_slot].p;
            memcpy(gat.pick_idx, q.where, (size_t)q.k_pick * 4);
            gat.k = (uint32_t)q.k_pick;
            gat.info = ra.info;
            gat.edits = ra.edits;
            gat.stride_e = ctx->p_stride_e;
            gat.out = h + PICK_RECORDS_AT;
            gat.out_stride = pick_record_stride(ctx);
            gat.flag = (uint32_t *)h;
            gat.seq = q.gather_seq;
            gat.arrived = (uint32_t *)ctx->d_done.p + MAX_CHAINS;
            if (reb_picks)
                for (int32_t j = 0; j < q.k_pick; j++)
                    rebuilt |= 1ull << q.pick_chain[j];
        }
        if (reb_ext)
        {
            ra.ext = (const RebuildExt *)(hx + q.ext_o_x);
            ra.ext_edits = (const lvbgpu_edit_dev *)(hx + q.ext_o_e);
            for (int32_t j = 0; j < q.k_ext; j++)
                rebuilt |= 1ull << ((const RebuildExt *)(hx + q.ext_o_x))[j].chain;
        }
    }
    if (gen)
    {
        pa.gen = *gen;
        uint32_t waits = 0;
        for (uint32_t i = 0; i < pa.gen.nseg; i++)
        {
            pa.gen.seg[i].wait = ((rebuilt >> pa.gen.seg[i].chain) & 1u) ? 1 : 0;
            waits += pa.gen.seg[i].wait;
        }
        if (waits)
        {
            const size_t old = ctx->d_table_ready.cap;
            HIPCHK(ctx, ctx->d_table_ready.reserve((size_t)MAX_CHAINS * 4));
            if (ctx->d_table_ready.cap != old)
                HIPCHK(ctx, hipMemsetAsync(ctx->d_table_ready.p, 0, ctx->d_table_ready.cap, ctx->stream));
            if (++ctx->post_seq == 0u)
                ctx->post_seq = 1u;
            pa.gen.table_ready = (const uint32_t *)ctx->d_table_ready.p;
            pa.gen.ready_seq = ctx->post_seq;
            pa.reb.table_ready = (uint32_t *)ctx->d_table_ready.p;
            pa.reb.ready_seq = ctx->post_seq;
            // test hook: the rebuilding workgroups keep their words to themselves and the waiting ones look only a few
            // thousand times - what a generating workgroup does when its wait runs out (its candidates become "not
            // proposals", nothing hangs) is otherwise never seen
            static const bool withhold = getenv("LVBGPU_DEBUG_WITHHOLD_READY") != nullptr;
            if (withhold)
            {
                pa.reb.withhold_ready = 1u;
                pa.gen.wait_spins = 4096u;
            }
        }
        ctx->post_launches_with_generator++;
    }
    static const bool post_profile = getenv("LVBGPU_POST_PROFILE") != nullptr;
    if (post_profile)
    {
        HIPCHK(ctx, ctx->d_post_prof.reserve((1 + 4 * 1000 + 64) * 8));
        HIPCHK(ctx, hipMemsetAsync(ctx->d_post_prof.p, 0, (1 + 4 * 1000 + 64) * 8, ctx->stream));
        pa.prof = (unsigned long long *)ctx->d_post_prof.p;
        pa.reb.prof = pa.prof + 1 + 4 * 1000;
    }
    ctx->post_launches++;
    // (beside another context's walk - lanes - the post launch takes twice as long: 36 -> 81-86 us at 32 chains in two
    // lanes.  Tried against that and not kept: 4-wave workgroups that fit wherever a walk workgroup retires, s_setprio for
    // its waves (together 1.3 % of the 32-chain run), a stream of the highest priority ordered against the main stream by
    // two events (1.216 -> 1.359 s).  profiles/experiments/r04_post_launch_and_lanes.md)
    HIPCHK(ctx, launch_post(pa, ctx->stream));
    return LVBGPU_OK;
}

// Accepted candidates of the batch in slot `src_slot` join what is pending; where[j] is the batch position of pick j.
int defer_picks(lvbgpu_ctx *ctx, int src_slot, int32_t k, const uint32_t *where, const int32_t *chains, uint64_t chain_mask,
                int *slot_out, uint32_t *seq_out, bool *tables_on_device_out)
{
    lvbgpu_ctx::Pending &pd = ctx->pend;
    if (pd.k_pick || (pd.chains & chain_mask)) // one batch's picks and one move per chain in a launch
    {
        const int rf = flush_pending(ctx, nullptr);
        if (rf != LVBGPU_OK)
            return rf;
    }
    int slot = 0;
    HIPCHK(ctx, take_pick_slot(ctx, &slot));
    HIPCHK(ctx, ctx->h_pick[slot].reserve(PICK_RECORDS_AT + (size_t)MAX_CHAINS * pick_record_stride(ctx)));
    // the generator's tables of the picked chains follow their moves on the device where they describe the trees the
    // candidates were drawn from (the picks were checked against the chains' versions by the caller)
    bool tables_on_device = (uint32_t)ctx->nb <= REBUILD_MAX_NODES && ctx->d_topo4.p != nullptr;
    for (int32_t j = 0; j < k && tables_on_device; j++)
        tables_on_device = ctx->parked[(size_t)chains[j]].d_topo_version == ctx->parked[(size_t)chains[j]].topo_version;
    pd.k_pick = k;
    pd.src_slot = src_slot;
    memcpy(pd.where, where, (size_t)k * 4);
    memcpy(pd.pick_chain, chains, (size_t)k * 4);
    pd.pick_rebuild = tables_on_device;
    pd.gather_slot = slot;
    pd.gather_seq = ++ctx->pick_seq;
    pd.chains |= chain_mask;
    *slot_out = slot;
    *seq_out = pd.gather_seq;
    *tables_on_device_out = tables_on_device;
    return LVBGPU_OK;
}
} // namespace lvbgpu_detail

extern "C" int lvbgpu_chains_commit(lvbgpu_ctx *ctx, int32_t k, const lvbgpu_chain_pick *picks)
{
    if (!ctx || k < 1 || k > MAX_CHAINS || !picks)
        return LVBGPU_E_ARG;
    lvbgpu_ctx::PropSlot &ps = ctx->pslot[ctx->last_slot];
    if (ps.segs.empty() || !ps.batch || ps.in_flight)
        return ctx->fail(LVBGPU_E_STATE, "no collected device batch to pick from: call lvbgpu_chains_propose_score first");
    ctx->last_pick_count = 0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    AllParked guard(ctx);
    {
        const int rf = resolve_follow(ctx); // the commit before this one: long done (there is one follow record)
        if (rf != LVBGPU_OK)
            return rf;
    }
    lvbgpu_batch *bt = ps.batch;
    // where each pick sits in the batch; nothing may have changed that chain's tree since it was drawn
    std::vector<uint32_t> where((size_t)k);
    uint64_t seen = 0;
    for (int32_t j = 0; j < k; j++)
    {
        const lvbgpu_chain_pick &pk = picks[j];
        const lvbgpu_ctx::PSeg *seg = nullptr;
        for (const lvbgpu_ctx::PSeg &sgm : ps.segs)
            if (sgm.chain == pk.chain)
                seg = &sgm;
        if (!seg || pk.b < 0 || pk.b >= seg->count || ((seen >> pk.chain) & 1u))
            return ctx->fail(LVBGPU_E_ARG, "pick " + std::to_string(j) + ": that chain drew no such candidate in the last batch, or is picked twice");
        if (seg->version != ctx->parked[(size_t)pk.chain].topo_version)
            return ctx->fail(LVBGPU_E_STATE, "the resident tree of chain " + std::to_string(pk.chain) + " changed since that batch was drawn");
        seen |= 1ull << pk.chain;
        where[(size_t)j] = (uint32_t)(seg->start + pk.b);
        if (((const int64_t *)bt->h_len.p)[where[(size_t)j]] >= PROPOSAL_OVERFLOW_LENGTH)
            return ctx->fail(LVBGPU_E_ARG, "pick " + std::to_string(j) + ": that candidate overflowed the per-candidate buffers");
    }
    std::vector<int32_t> pchains((size_t)k);
    for (int32_t j = 0; j < k; j++)
        pchains[(size_t)j] = picks[j].chain;
    int slot = 0;
    uint32_t seq = 0;
    bool tables_on_device = false;
    {
        const int ra = defer_picks(ctx, ctx->last_slot, k, where.data(), pchains.data(), seen, &slot, &seq, &tables_on_device);
        if (ra != LVBGPU_OK)
            return ra;
    }
    // 3. the host's topologies follow LATER (ctx->follow): the records are on their way into the pinned slot; versions
    //    move now, so that everything that compares versions (stale picks, stale batches) sees the tree as changed
    ctx->follow.pending = true;
    ctx->follow.slot = slot;
    ctx->follow.k = k;
    ctx->follow.seq = seq;
    for (int32_t j = 0; j < k; j++)
    {
        ctx->follow.chains[j] = picks[j].chain;
        ctx->follow.has[j] = true;
        ChainSlot &cs = ctx->parked[(size_t)picks[j].chain];
        cs.topo_version = ++ctx->version_counter;
        cs.cur_length_stale = true;
        if (tables_on_device)
            cs.d_topo_version = cs.topo_version; // rebuilt in place by the post launch
    }
    forget_named_candidates(ctx, seen);
    ctx->last_pick_count = k;
    for (int32_t j = 0; j < k; j++)
        ctx->last_pick_has[j] = true;
    return LVBGPU_OK;
}

// ---- a whole step: submit with rules, collect with picks (include/lvbgpu.h) ------------------------------------------
extern "C" int lvbgpu_chains_step_submit(lvbgpu_ctx *ctx, int32_t slot, int32_t k, const lvbgpu_chain_draw *draws,
                                         const lvbgpu_chain_rule *rules)
{
    if (!rules)
        return LVBGPU_E_ARG;
    // (the host side of the step before this one - its accepted moves' way into the host topologies - is NOT waited for
    // here: the moves may still be on their way, and on the accept path the host is what the device waits for; the
    // collect settles it before it records this step's)
    return propose_submit(ctx, slot, k, draws, nullptr, rules);
}

extern "C" int lvbgpu_chains_step_collect(lvbgpu_ctx *ctx, int32_t slot, int64_t *lengths_out, int32_t *picks_out)
{
    if (!ctx || slot < 0 || slot >= lvbgpu_ctx::PROP_SLOTS || !picks_out)
        return LVBGPU_E_ARG;
    lvbgpu_ctx::PropSlot &ps = ctx->pslot[slot];
    if (!ps.in_flight || !ps.step)
        return ctx->fail(LVBGPU_E_STATE, "no step was submitted in that slot");
    int rc = propose_collect(ctx, slot, lengths_out);
    const int32_t k = (int32_t)ps.host_rules.size();
    ps.step = false;
    if (rc != LVBGPU_OK)
        return rc; // (nothing was committed: the chains' trees stand as they were)
    ctx->step_map.assign((size_t)k, -1);
    // the library decides, by the one function (decide.h), and commits as lvbgpu_chains_commit does
    std::vector<lvbgpu_chain_pick> picks;
    for (int32_t i = 0; i < k; i++)
    {
        const DecideRule &r = ps.host_rules[(size_t)i];
        picks_out[i] = -1;
        for (uint32_t j = 0; j < r.count; j++)
            if (lvb_take(lengths_out[r.start + j] == INT64_MAX ? (long long)LVB_OVERFLOW_LENGTH : (long long)lengths_out[r.start + j], &r, j))
            {
                picks_out[i] = (int32_t)j;
                ctx->step_map[(size_t)i] = (int32_t)picks.size();
                picks.push_back({ps.segs[(size_t)i].chain, (int32_t)j});
                break;
            }
    }
    return picks.empty() ? LVBGPU_OK : lvbgpu_chains_commit(ctx, (int32_t)picks.size(), picks.data());
}

extern "C" int lvbgpu_chains_step_edits(lvbgpu_ctx *ctx, int32_t i, lvbgpu_edit *edits, int32_t cap, int32_t *n_edits)
{
    if (!ctx || i < 0)
        return LVBGPU_E_ARG;
    if ((size_t)i >= ctx->step_map.size() || ctx->step_map[(size_t)i] < 0)
        return ctx->fail(LVBGPU_E_ARG, "that draw accepted nothing in the last step (or there was none)");
    return lvbgpu_chains_picked_edits(ctx, ctx->step_map[(size_t)i], edits, cap, n_edits);
}

// Re-root several chains in ONE commit walk (arbreroot, TreeOperations.c:639-656, as rewrites along the old-root ..
// new-root path): what lvbgpu_select_chain + lvbgpu_commit(edits, new_root, NULL) does chain by chain - R chains
// re-rooting every 1000 proposals each were a third of a step's device time that way.  The generator's tables follow
// on the device.  Asynchronous: a re-root does not change the length.
namespace lvbgpu_detail
{
// Rewrites of several chains' resident trees (distinct chains; a new root leaf per chain or -1) walked in ONE commit walk,
// the generator's tables following on the device: the common part of lvbgpu_chains_reroot and
// lvbgpu_chains_commit_edits.  The caller holds AllParked and has resolved the pending follow.
int commit_rewrites(lvbgpu_ctx *ctx, int32_t k, const int32_t *chains, std::vector<std::vector<Edit>> &edits, const int32_t *new_roots,
                    uint64_t seen, const char *what, bool rebuild_tables)
{
    Packed pk;
    Program prog;
    for (int32_t j = 0; j < k; j++)
    {
        ChainSlot &cs = ctx->parked[(size_t)chains[j]];
        std::vector<Edit> &ed = edits[(size_t)j];
        const size_t tok0 = prog.toks.size(), dst0 = prog.dsts.size();
        std::string why;
        prog.max_stack = 0;
        if (!ctx->pb.build_candidate(cs.topo, ed.data(), (int32_t)ed.size(), new_roots[j], prog, &why))
            return ctx->fail(LVBGPU_E_TOPOLOGY, std::string(what) + " " + std::to_string(j) + ": " + why);
        pk.add(prog, tok0, dst0, 0, (uint32_t)chains[j] << CAND_CHAIN_SHIFT);
        pk.max_stack = std::max(pk.max_stack, prog.max_stack);
    }
    int rc = check_depth(ctx, pk.max_stack);
    if (rc != LVBGPU_OK)
        return rc;
    // one move per chain and one block of host-built programs in a launch
    if (ctx->pend.k_ext || (ctx->pend.chains & seen))
    {
        rc = flush_pending(ctx, nullptr);
        if (rc != LVBGPU_OK)
            return rc;
    }
    // one pinned slot: [descriptors][tokens][destinations][what the table rebuild needs][all rewrites]
    size_t n_all_edits = 0;
    for (const auto &e : edits)
        n_all_edits += e.size();
    const size_t o_t = align16((size_t)k * sizeof(CandDesc));
    const size_t o_d = o_t + align16(prog.toks.size() * 4);
    const size_t o_x = o_d + align16(prog.dsts.size() * 4);
    const size_t o_e = o_x + align16((size_t)k * sizeof(RebuildExt));
    const size_t total = o_e + align16(n_all_edits * sizeof(lvbgpu_edit_dev));
    int slot = 0;
    HIPCHK(ctx, take_pick_slot(ctx, &slot));
    const size_t slot_min = 64 + align16((size_t)MAX_CHAINS * 4) + (size_t)MAX_CHAINS * 16; // never below what a pick needs first
    HIPCHK(ctx, ctx->h_pick[slot].reserve(std::max(total, slot_min)));
    char *h = (char *)ctx->h_pick[slot].p;
    memcpy(h, pk.cands.data(), (size_t)k * sizeof(CandDesc));
    memcpy(h + o_t, prog.toks.data(), prog.toks.size() * 4);
    memcpy(h + o_d, prog.dsts.data(), prog.dsts.size() * 4);
    RebuildExt *ext = (RebuildExt *)(h + o_x);
    lvbgpu_edit_dev *all = (lvbgpu_edit_dev *)(h + o_e);
    size_t off = 0;
    for (int32_t j = 0; j < k; j++)
    {
        ext[j] = {chains[j], new_roots[j] >= 0 ? new_roots[j] : ctx->parked[(size_t)chains[j]].topo.root, (int32_t)off, (int32_t)edits[(size_t)j].size()};
        memcpy(all + off, edits[(size_t)j].data(), edits[(size_t)j].size() * sizeof(lvbgpu_edit_dev));
        off += edits[(size_t)j].size();
    }
    static_assert(sizeof(Edit) == sizeof(lvbgpu_edit_dev), "edit layout");
    // the tables follow on the device when they describe the trees as they are now
    // (rebuild_tables false: the caller draws on the host for now - the tables go stale and are made again, on the host,
    // when a device draw next needs them)
    bool tables_on_device = rebuild_tables && (uint32_t)ctx->nb <= REBUILD_MAX_NODES && ctx->d_topo4.p != nullptr;
    for (int32_t j = 0; j < k && tables_on_device; j++)
        tables_on_device = ctx->parked[(size_t)chains[j]].d_topo_version == ctx->parked[(size_t)chains[j]].topo_version;
    // the device gets all of it with the next post launch (flush_pending): together with the accepted device moves of
    // other chains, and with the next step's generator if a submit comes next
    {
        lvbgpu_ctx::Pending &pd = ctx->pend;
        pd.k_ext = k;
        pd.ext_slot = slot;
        pd.ext_o_t = o_t;
        pd.ext_o_d = o_d;
        pd.ext_o_x = o_x;
        pd.ext_o_e = o_e;
        pd.ext_max_stack = std::max(pk.max_stack, 1);
        pd.ext_rebuild = tables_on_device;
        pd.chains |= seen;
    }
    // the host's topologies follow
    for (int32_t j = 0; j < k; j++)
    {
        ChainSlot &cs = ctx->parked[(size_t)chains[j]];
        std::string why;
        if (!ctx->pb.apply_edits(cs.topo, edits[(size_t)j].data(), (int32_t)edits[(size_t)j].size(), new_roots[j], &why))
        {
            cs.have_tree = false;
            return ctx->fail(LVBGPU_E_TOPOLOGY, why);
        }
        cs.topo_version = ++ctx->version_counter;
        cs.cur_length_stale = true;
        if (tables_on_device)
            cs.d_topo_version = cs.topo_version;
    }
    forget_named_candidates(ctx, seen);
    return LVBGPU_OK;
}
} // namespace lvbgpu_detail

extern "C" int lvbgpu_chains_reroot(lvbgpu_ctx *ctx, int32_t k, const lvbgpu_chain_root *reqs)
{
    if (!ctx || k < 1 || k > MAX_CHAINS || !reqs)
        return LVBGPU_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    AllParked guard(ctx);
    // the programs below are built from the chains' topologies as they are NOW: a chain whose accepted move is still on its
    // way to the host needs it first (and with it the post launch that sends it).  Other chains' moves may stay pending -
    // in an annealing step the accepted moves and the re-roots are different chains nearly always, and go out together
    if (ctx->follow.pending)
    {
        bool mine = false;
        for (int32_t j = 0; j < k && !mine; j++)
            for (int32_t i = 0; i < ctx->follow.k; i++)
                mine |= ctx->follow.has[i] && ctx->follow.chains[i] == reqs[j].chain;
        if (mine)
        {
            const int rf = resolve_follow(ctx);
            if (rf != LVBGPU_OK)
                return rf;
        }
    }
    uint64_t seen = 0;
    std::vector<std::vector<Edit>> edits((size_t)k);
    int32_t chains[MAX_CHAINS], new_roots[MAX_CHAINS];
    for (int32_t j = 0; j < k; j++)
    {
        const lvbgpu_chain_root &rq = reqs[j];
        if (rq.chain < 0 || rq.chain >= ctx->nchains || ((seen >> rq.chain) & 1u))
            return ctx->fail(LVBGPU_E_ARG, "re-root " + std::to_string(j) + ": chain out of range or listed twice");
        seen |= 1ull << rq.chain;
        ChainSlot &cs = ctx->parked[(size_t)rq.chain];
        if (!cs.have_tree)
            return ctx->fail(LVBGPU_E_STATE, "no resident tree: call lvbgpu_set_tree first");
        const Topology &t = cs.topo;
        if (rq.new_root < 0 || rq.new_root >= t.n || rq.new_root == t.root)
            return ctx->fail(LVBGPU_E_TOPOLOGY, "re-root " + std::to_string(j) + ": the new root must be another leaf");
        // every node on the way up takes (its old parent, its old sister); the old root becomes an ordinary leaf
        std::vector<Edit> &ed = edits[(size_t)j];
        for (int32_t c = rq.new_root; c != t.root; c = t.parent[c])
        {
            const int32_t p = t.parent[c];
            ed.push_back({c, p, t.left[p] == c ? t.right[p] : t.left[p]});
        }
        ed.push_back({t.root, UNSET, UNSET});
        chains[j] = rq.chain;
        new_roots[j] = rq.new_root;
    }
    return commit_rewrites(ctx, k, chains, edits, new_roots, seen, "re-root", true);
}

// Accept host-made candidates of several chains at once: chain chains[j]'s resident tree takes the rewrites
// edits[edit_offsets[j] .. edit_offsets[j + 1]) - any set of child-pair rewrites that gives a tree, e.g. the cumulative
// rewrites of a run of accepted moves - in ONE commit walk for all of them; what lvbgpu_select_chain + lvbgpu_commit(.., NULL)
// does chain by chain.  Asynchronous: the caller scored the candidates and knows the lengths.
extern "C" int lvbgpu_chains_commit_edits(lvbgpu_ctx *ctx, int32_t k, const int32_t *chains, const int32_t *edit_offsets,
                                          const lvbgpu_edit *edits_in)
{
    if (!ctx || k < 1 || k > MAX_CHAINS || !chains || !edit_offsets || !edits_in)
        return LVBGPU_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    AllParked guard(ctx);
    {
        const int rf = resolve_follow(ctx);
        if (rf != LVBGPU_OK)
            return rf;
    }
    uint64_t seen = 0;
    std::vector<std::vector<Edit>> edits((size_t)k);
    int32_t new_roots[MAX_CHAINS];
    for (int32_t j = 0; j < k; j++)
    {
        if (chains[j] < 0 || chains[j] >= ctx->nchains || ((seen >> chains[j]) & 1u))
            return ctx->fail(LVBGPU_E_ARG, "commit " + std::to_string(j) + ": chain out of range or listed twice");
        seen |= 1ull << chains[j];
        if (!ctx->parked[(size_t)chains[j]].have_tree)
            return ctx->fail(LVBGPU_E_STATE, "no resident tree: call lvbgpu_set_tree first");
        if (edit_offsets[j + 1] <= edit_offsets[j])
            return ctx->fail(LVBGPU_E_ARG, "commit " + std::to_string(j) + ": no rewrites (or edit_offsets not monotone)");
        const Edit *e = reinterpret_cast<const Edit *>(edits_in) + edit_offsets[j];
        edits[(size_t)j].assign(e, e + (edit_offsets[j + 1] - edit_offsets[j]));
        new_roots[j] = -1;
    }
    // Every one of them a candidate of the last lvbgpu_chains_score_edits call, its chain's tree unchanged since?  Then
    // their programs lie in that batch: one commit walk over them, picked by position (as lvbgpu_chains_commit does with
    // device-built batches) - nothing is built and nothing uploaded.  (k commit programs built one after the other were
    // most of a host-drawn step of many chains.)
    uint32_t where[MAX_CHAINS];
    lvbgpu_batch *sb = ctx->scored_batch;
    bool reuse = sb && sb == ctx->step_batch[0] && sb->build_gen == ctx->scored_gen && !sb->in_place;
    for (int32_t j = 0; j < k && reuse; j++)
    {
        const int32_t n = edit_offsets[j + 1] - edit_offsets[j];
        const uint64_t h = edits_hash(edits_in + edit_offsets[j], n);
        const uint64_t v = ctx->parked[(size_t)chains[j]].topo_version;
        bool found = false;
        for (size_t i = 0; i < ctx->scored.size() && !found; i++)
        {
            const lvbgpu_ctx::ScoredEdit &se = ctx->scored[i];
            // (found by the hash, told apart by the rewrites themselves - as the treestack does with its trees)
            if (se.chain == chains[j] && se.n_edits == n && se.hash == h && se.version == v &&
                memcmp(ctx->scored_edits.data() + se.edit_off, edits_in + edit_offsets[j], (size_t)n * sizeof(lvbgpu_edit)) == 0)
            {
                where[j] = (uint32_t)i;
                found = true;
            }
        }
        reuse = found;
    }
    if (!reuse)
        return commit_rewrites(ctx, k, chains, edits, new_roots, seen, "commit", false);
    {
        const int rp = flush_pending(ctx, nullptr); // (this walk is launched here and now)
        if (rp != LVBGPU_OK)
            return rp;
    }
    {
        const size_t old_done = ctx->d_done.cap;
        HIPCHK(ctx, ctx->d_done.reserve((size_t)(MAX_CHAINS + 1) * 4));
        if (ctx->d_done.cap != old_done) // once per context
        {
            HIPCHK(ctx, hipMemsetAsync(ctx->d_done.p, 0, ctx->d_done.cap, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    HIPCHK(ctx, ctx->d_tmp_changes.reserve((size_t)MAX_CHAINS * (size_t)(ctx->nb + 1) * 8));
    if (ctx->tmp_changes_zeroed_cap != ctx->d_tmp_changes.cap)
    {
        HIPCHK(ctx, hipMemsetAsync(ctx->d_tmp_changes.p, 0, ctx->d_tmp_changes.cap, ctx->stream));
        ctx->tmp_changes_zeroed_cap = ctx->d_tmp_changes.cap;
    }
    HIPCHK(ctx, ctx->d_len.reserve(8));
    WalkArgs a = resident_args(ctx, sb->d_prog.p, sb->off_toks, sb->off_dsts, ctx->d_len.p, (uint32_t)k, (int32_t)sb->stats.max_stack);
    a.use_pick = 1;
    memcpy(a.pick_idx, where, (size_t)k * 4);
    a.s_all_out = (unsigned long long *)ctx->d_scalars;
    a.tmp_changes = (unsigned long long *)ctx->d_tmp_changes.p;
    a.tmp_stride = (uint32_t)(ctx->nb + 1);
    a.done_count = (uint32_t *)ctx->d_done.p;
    HIPCHK(ctx, launch_walk(a, true, ctx->stream));
    // the host's topologies follow (every chain on its own: on the pool's threads from a few chains on); the
    // generator's tables go stale
    {
        std::vector<std::string> whys((size_t)k);
        std::vector<uint8_t> ok((size_t)k, 0);
        auto apply_one = [&](int32_t j, ProgramBuilder &pb) {
            ChainSlot &cs = ctx->parked[(size_t)chains[j]];
            ok[(size_t)j] = pb.apply_edits(cs.topo, edits[(size_t)j].data(), (int32_t)edits[(size_t)j].size(), -1, &whys[(size_t)j]) ? 1 : 0;
        };
        const int T = (ctx->pool && k >= 8) ? std::min(ctx->pool->size(), (int)k / 2) : 1;
        if (T > 1)
        {
            if ((int)ctx->workers.size() < T)
                ctx->workers.resize((size_t)T);
            ctx->pool->run(T, [&](int t) {
                BuildWorker &w = ctx->workers[(size_t)t];
                w.pb.resize(ctx->nb);
                w.topo_version = ~0ull; // (its builder's scratch is used, its topology copy is not)
                for (int32_t j = (int32_t)((int64_t)k * t / T), e = (int32_t)((int64_t)k * (t + 1) / T); j < e; j++)
                    apply_one(j, w.pb);
            });
        }
        else
            for (int32_t j = 0; j < k; j++)
                apply_one(j, ctx->pb);
        for (int32_t j = 0; j < k; j++)
        {
            ChainSlot &cs = ctx->parked[(size_t)chains[j]];
            if (!ok[(size_t)j])
            {
                cs.have_tree = false;
                return ctx->fail(LVBGPU_E_TOPOLOGY, whys[(size_t)j]);
            }
            cs.topo_version = ++ctx->version_counter;
            cs.cur_length_stale = true;
        }
    }
    forget_named_candidates(ctx, seen);
    ctx->commits_reusing_programs++;
    return LVBGPU_OK;
}

extern "C" int lvbgpu_chains_picked_edits(lvbgpu_ctx *ctx, int32_t j, lvbgpu_edit *edits, int32_t cap, int32_t *n_edits)
{
    if (!ctx || !edits || !n_edits || j < 0)
        return LVBGPU_E_ARG;
    if (j >= ctx->last_pick_count)
        return ctx->fail(LVBGPU_E_STATE, "the last lvbgpu_chains_commit had no such pick");
    {
        const int rf = settle(ctx); // waits for the records if they are still on their way
        if (rf != LVBGPU_OK)
            return rf;
    }
    if (!ctx->last_pick_has[j])
        return ctx->fail(LVBGPU_E_ARG, "that chain accepted nothing in the last step");
    if ((size_t)(j + 1) * ctx->pick_records_stride > ctx->pick_records.size())
        return ctx->fail(LVBGPU_E_STATE, "the picked moves never arrived from the device");
    const char *rec = ctx->pick_records.data() + (size_t)j * ctx->pick_records_stride; // (resolve_follow's copy)
    const ProposalInfo pi = *(const ProposalInfo *)rec;
    if (pi.n_edits > cap)
        return ctx->fail(LVBGPU_E_ARG, "edit buffer too small");
    memcpy(edits, rec + sizeof(ProposalInfo), (size_t)pi.n_edits * sizeof(lvbgpu_edit));
    *n_edits = pi.n_edits;
    return LVBGPU_OK;
}

extern "C" int lvbgpu_proposal_edits(lvbgpu_ctx *ctx, int32_t b, lvbgpu_edit *edits, int32_t cap, int32_t *n_edits,
                                     int32_t *info4)
{
    if (!ctx || !edits || !n_edits)
        return LVBGPU_E_ARG;
    lvbgpu_ctx::PropSlot &ps = ctx->pslot[0];
    if (ps.p_B <= 0 || ps.in_flight || b < 0 || b >= ps.p_B)
        return ctx->fail(LVBGPU_E_STATE, "no device batch holds that candidate: call lvbgpu_propose_score first");
    if (ctx->d_topo_version != ctx->topo_version)
        return ctx->fail(LVBGPU_E_STATE, "the resident tree changed since that batch was drawn");
    ENTER(ctx);
    // this candidate's descriptor, then its edits
    HIPCHK(ctx, ctx->h_pinfo.reserve(sizeof(ProposalInfo)));
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinfo.p, (const ProposalInfo *)ps.d_pinfo.p + b, sizeof(ProposalInfo),
                               hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const ProposalInfo pi = *(const ProposalInfo *)ctx->h_pinfo.p;
    if (pi.overflow)
        return ctx->fail(LVBGPU_E_ARG, "that candidate overflowed the per-candidate buffers");
    if (pi.n_edits > cap)
        return ctx->fail(LVBGPU_E_ARG, "edit buffer too small");
    static_assert(sizeof(lvbgpu_edit) == sizeof(lvbgpu_edit_dev), "edit layout");
    HIPCHK(ctx, hipMemcpyAsync(edits, (const lvbgpu_edit_dev *)ps.d_pedits.p + (size_t)b * ctx->p_stride_e,
                               (size_t)pi.n_edits * sizeof(lvbgpu_edit), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *n_edits = pi.n_edits;
    if (info4)
    {
        info4[0] = pi.kind;
        info4[1] = pi.a;
        info4[2] = pi.kind == 0 ? pi.flag : pi.b;
        info4[3] = pi.c;
    }
    return LVBGPU_OK;
}


// what the last device-built batch cost (the counts lvbgpu_batch_get_stats gives for host-built ones): the
// candidates' descriptors are read back and summed - for the measurement line, not for the search
extern "C" int lvbgpu_proposal_stats(lvbgpu_ctx *ctx, lvbgpu_batch_stats *out)
{
    if (!ctx || !out)
        return LVBGPU_E_ARG;
    lvbgpu_ctx::PropSlot &ps = ctx->pslot[ctx->last_slot];
    int64_t total = 0;
    for (const lvbgpu_ctx::PSeg &sgm : ps.segs)
        total += sgm.count;
    if (total <= 0 || !ps.batch || ps.in_flight)
        return ctx->fail(LVBGPU_E_STATE, "no device batch: call lvbgpu_propose_score first");
    ENTER(ctx);
    std::vector<ProposalInfo> info((size_t)total);
    HIPCHK(ctx, hipMemcpyAsync(info.data(), ps.d_pinfo.p, info.size() * sizeof(ProposalInfo), hipMemcpyDeviceToHost,
                               ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    lvbgpu_batch_stats st{};
    st.max_stack = 1;
    for (const ProposalInfo &pi : info)
    {
        if (pi.overflow)
            continue;
        st.candidates++;
        st.combines += pi.ncomb;       // D + 2
        st.rows_read += pi.ncomb + 1;  // D + 3 clean rows
        st.dirty_nodes += pi.ncomb - 2;
    }
    st.algorithmic_bytes = st.rows_read * ctx->nwords * 8;
    *out = st;
    return LVBGPU_OK;
}
