// api_propose.cpp - liblvbgpu.so: neighbourhoods whose rewrites and programs are built on the device (drawn there, or named by the host).
#include "ctx.hpp"

// ---- device-side neighbourhoods

namespace lvbgpu_detail
{
// the generator's tables of the resident topology (layout: GenArgs in kernels.hpp)
template <typename IdxT>
static void fill_tables(const Topology &t, int32_t K, const std::vector<int32_t> &order, const std::vector<int32_t> &depth,
                        const std::vector<int32_t> &nleaf, std::vector<IdxT> &out)
{
    const size_t nb = (size_t)t.nb;
    const size_t nlo = (size_t)t.n; // leaf_order: n - 1 used
    out.assign((7 + (size_t)K) * nb + nlo + 8, (IdxT)0);
    IdxT *parent = out.data(), *left = parent + nb, *right = left + nb, *nl = right + nb, *dep = nl + nb, *tin = dep + nb,
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
}

int sync_device_topology(lvbgpu_ctx *ctx)
{
    if (ctx->d_topo_version == ctx->topo_version)
        return LVBGPU_OK;
    const int32_t nb = ctx->nb;
    const Topology &t = ctx->topo;
    // preorder from the root leaf; children before parents when read backwards
    std::vector<int32_t> order;
    order.reserve(nb);
    std::vector<int32_t> depth((size_t)nb, 0), nleaf((size_t)nb, 1);
    std::vector<int32_t> st{t.root};
    int32_t maxdepth = 1;
    while (!st.empty())
    {
        const int32_t v = st.back();
        st.pop_back();
        order.push_back(v);
        if (t.left[v] >= 0)
        {
            depth[t.left[v]] = depth[t.right[v]] = depth[v] + 1;
            maxdepth = std::max(maxdepth, depth[v] + 1);
            st.push_back(t.right[v]);
            st.push_back(t.left[v]);
        }
    }
    for (auto it = order.rbegin(); it != order.rend(); ++it)
        if (t.left[*it] >= 0)
            nleaf[*it] = nleaf[t.left[*it]] + nleaf[t.right[*it]];
    int32_t K = 1;
    while ((1 << K) <= maxdepth)
        K++;
    size_t bytes;
    const void *src;
    if (nb <= 65535)
    {
        fill_tables<uint16_t>(t, K, order, depth, nleaf, ctx->gen_tab16);
        bytes = ctx->gen_tab16.size() * 2;
        src = ctx->gen_tab16.data();
        ctx->gen_idx_bytes = 2;
    }
    else
    {
        fill_tables<int32_t>(t, K, order, depth, nleaf, ctx->gen_tab32);
        bytes = ctx->gen_tab32.size() * 4;
        src = ctx->gen_tab32.data();
        ctx->gen_idx_bytes = 4;
    }
    bytes &= ~(size_t)15; // whole 16-byte pieces (the arrays end 8 elements before the vector does)
    // one slot per chain, wide enough for the deepest tree these taxa can form (K <= bits of 2n-3)
    {
        int32_t kmax = 1;
        while ((1 << kmax) <= nb)
            kmax++;
        const size_t widest = ((7 + (size_t)kmax) * (size_t)nb + (size_t)ctx->n + 8) * ctx->gen_idx_bytes;
        ctx->gen_table_stride = (uint32_t)((widest + 255) & ~(size_t)255);
    }
    if (bytes > ctx->gen_table_stride)
        return ctx->fail(LVBGPU_E_ARG, "generator tables exceed their slot");
    const size_t old_cap = ctx->d_topo4.cap;
    HIPCHK(ctx, ctx->d_topo4.reserve((size_t)ctx->nchains * ctx->gen_table_stride));
    if (ctx->d_topo4.cap != old_cap) // a new buffer holds no chain's tables
        for (ChainSlot &cs : ctx->parked)
            cs.d_topo_version = ~0ull;
    HIPCHK(ctx, ctx->h_topo.reserve(bytes));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); // an upload still reading the staging buffer
    memcpy(ctx->h_topo.p, src, bytes);
    HIPCHK(ctx, hipMemcpyAsync((char *)ctx->d_topo4.p + (size_t)ctx->chain * ctx->gen_table_stride, ctx->h_topo.p, bytes,
                               hipMemcpyHostToDevice, ctx->stream));
    ctx->gen_table_bytes = (uint32_t)bytes;
    ctx->gen_K = K;
    ctx->d_topo_version = ctx->topo_version;
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

int propose_score_impl(lvbgpu_ctx *ctx, int32_t B, int32_t kind, uint32_t mix_a, uint32_t mix_b, uint64_t seed,
                       int64_t *lengths_out, const lvbgpu_move *moves = nullptr)
{
    if (!ctx || B < 1 || kind < -3 || kind > 2 || !lengths_out)
        return LVBGPU_E_ARG;
    if (!ctx->have_tree)
        return ctx->fail(LVBGPU_E_STATE, "no resident tree: call lvbgpu_set_tree first");
    if (ctx->n < 5)
        return ctx->fail(LVBGPU_E_ARG, "rearrangements need at least 5 taxa");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = sync_device_topology(ctx);
    if (rc != LVBGPU_OK)
        return rc;
    // fixed strides: a program has at most (n-3)+3 tokens; edits are capped (longer TBR paths overflow)
    const uint32_t stride_t = (uint32_t)ctx->n + 8u;
    const uint32_t stride_e = (uint32_t)std::min<int64_t>(ctx->nb, 512);
    if ((uint64_t)B * stride_t >= (1ull << 32) || (uint64_t)B * ctx->ntiles >= (1ull << 31))
        return ctx->fail(LVBGPU_E_ARG, "batch too large");
    if (!ctx->prop_batch)
    {
        ctx->prop_batch = new (std::nothrow) lvbgpu_batch();
        if (!ctx->prop_batch)
            return LVBGPU_E_NOMEM;
    }
    lvbgpu_batch *bt = ctx->prop_batch;
    const size_t o_t = align16((size_t)B * sizeof(CandDesc));
    const size_t o_d = o_t + align16((size_t)B * stride_t * 4);
    const size_t total = o_d + align16((size_t)B * stride_t * 4);
    HIPCHK(ctx, bt->d_prog.reserve(total));
    // a buffer that grew holds whatever its new memory held - and may well sit at the old address, so it is
    // the capacity that tells, not the pointer
    const size_t old_len_cap = bt->d_len.cap;
    HIPCHK(ctx, bt->d_len.reserve((size_t)B * 8));
    if (bt->d_len.cap != old_len_cap)
        bt->len_zeroed = false;
    HIPCHK(ctx, bt->h_len.reserve((size_t)B * 8));
    HIPCHK(ctx, ctx->d_pedits.reserve((size_t)B * stride_e * sizeof(lvbgpu_edit_dev)));
    HIPCHK(ctx, ctx->d_pinfo.reserve((size_t)B * sizeof(ProposalInfo)));
    bt->ctx = ctx;
    bt->B = B;
    bt->off_toks = o_t;
    bt->off_dsts = o_d;
    bt->full_mode = false;
    bt->topo_version = ctx->topo_version;
    bt->chain = ctx->chain;
    bt->stats = lvbgpu_batch_stats{};
    bt->stats.candidates = B;
    bt->stats.max_stack = 1; // at most one sibling set waits while the other path is walked
    ctx->p_stride_t = stride_t;
    ctx->p_stride_e = stride_e;
    ctx->p_B = 0;
    const lvbgpu_move_dev *d_moves = nullptr;
    if (moves)
    {
        static_assert(sizeof(lvbgpu_move) == sizeof(lvbgpu_move_dev), "move layout");
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
                if (move_defect(ctx->topo, moves[b]))
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
                return ctx->fail(LVBGPU_E_TOPOLOGY, "move " + std::to_string(b) + ": " + move_defect(ctx->topo, moves[b]));
            }
        HIPCHK(ctx, ctx->d_moves.reserve((size_t)B * sizeof(lvbgpu_move)));
        HIPCHK(ctx, ctx->h_moves.reserve((size_t)B * sizeof(lvbgpu_move)));
        memcpy(ctx->h_moves.p, moves, (size_t)B * sizeof(lvbgpu_move)); // pinned staging: the caller's array may go away
        // a short list is read by the generator where it lies (16 bytes per thread, once); a long one is copied
        if (ctx->direct_steps && (size_t)B * sizeof(lvbgpu_move) <= DIRECT_READ_MAX_BYTES)
            d_moves = (const lvbgpu_move_dev *)ctx->h_moves.p;
        else
        {
            HIPCHK(ctx, hipMemcpyAsync(ctx->d_moves.p, ctx->h_moves.p, (size_t)B * sizeof(lvbgpu_move),
                                       hipMemcpyHostToDevice, ctx->stream));
            d_moves = (const lvbgpu_move_dev *)ctx->d_moves.p;
        }
    }
    GenArgs ga{};
    ga.tables = (const char *)ctx->d_topo4.p + (size_t)ctx->chain * ctx->gen_table_stride;
    ga.chain = (uint32_t)ctx->chain;
    ga.table_bytes = ctx->gen_table_bytes;
    ga.idx_bytes = ctx->gen_idx_bytes;
    ga.n = (int32_t)ctx->n;
    ga.nb = ctx->nb;
    ga.root = ctx->topo.root;
    ga.K = ctx->gen_K;
    ga.leaf_order_len = (uint32_t)ctx->n;
    ga.kind_all = kind;
    ga.mix_a = mix_a;
    ga.mix_b = mix_b;
    ga.seed = seed;
    ga.B = (uint32_t)B;
    ga.stride_t = stride_t;
    ga.stride_e = stride_e;
    ga.toks = (uint32_t *)((char *)bt->d_prog.p + o_t);
    ga.dsts = (int32_t *)((char *)bt->d_prog.p + o_d);
    ga.edits = (lvbgpu_edit_dev *)ctx->d_pedits.p;
    ga.cands = (CandDesc *)bt->d_prog.p;
    ga.info = (ProposalInfo *)ctx->d_pinfo.p;
    ga.len_out = (unsigned long long *)bt->d_len.p;
    ga.moves = d_moves;
    HIPCHK(ctx, launch_propose(ga, ctx->stream));
    bt->len_zeroed = true; // by the generator
    rc = lvbgpu_batch_launch(ctx, bt);
    if (rc != LVBGPU_OK)
        return rc;
    // only the lengths come back per step; a move's descriptor and edits are fetched when (and only
    // when) the caller wants that candidate (lvbgpu_proposal_edits)
    HIPCHK(ctx, hipMemcpyAsync(bt->h_len.p, bt->d_len.p, (size_t)B * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, wait_for_step(ctx, B));
    const int64_t *len = (const int64_t *)bt->h_len.p;
    for (int32_t b = 0; b < B; b++)
    {
        if (len[b] >= PROPOSAL_OVERFLOW_LENGTH)
        {
            lengths_out[b] = INT64_MAX;
            continue;
        }
        lengths_out[b] = len[b];
        if (len[b] <= 0)
            return ctx->fail(LVBGPU_E_ZEROLEN, "assertion failed: changes > 0 (device-built candidate " + std::to_string(b) +
                                                   " of " + std::to_string(B) + " scored " + std::to_string(len[b]) + ")");
    }
    ctx->p_B = B;
    return LVBGPU_OK;
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

extern "C" int lvbgpu_proposal_edits(lvbgpu_ctx *ctx, int32_t b, lvbgpu_edit *edits, int32_t cap, int32_t *n_edits,
                                     int32_t *info4)
{
    if (!ctx || !edits || !n_edits)
        return LVBGPU_E_ARG;
    if (ctx->p_B <= 0 || b < 0 || b >= ctx->p_B)
        return ctx->fail(LVBGPU_E_STATE, "no device batch holds that candidate: call lvbgpu_propose_score first");
    if (ctx->d_topo_version != ctx->topo_version)
        return ctx->fail(LVBGPU_E_STATE, "the resident tree changed since that batch was drawn");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // this candidate's descriptor, then its edits
    HIPCHK(ctx, ctx->h_pinfo.reserve(sizeof(ProposalInfo)));
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinfo.p, (const ProposalInfo *)ctx->d_pinfo.p + b, sizeof(ProposalInfo),
                               hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const ProposalInfo pi = *(const ProposalInfo *)ctx->h_pinfo.p;
    if (pi.overflow)
        return ctx->fail(LVBGPU_E_ARG, "that candidate overflowed the per-candidate buffers");
    if (pi.n_edits > cap)
        return ctx->fail(LVBGPU_E_ARG, "edit buffer too small");
    static_assert(sizeof(lvbgpu_edit) == sizeof(lvbgpu_edit_dev), "edit layout");
    HIPCHK(ctx, hipMemcpyAsync(edits, (const lvbgpu_edit_dev *)ctx->d_pedits.p + (size_t)b * ctx->p_stride_e,
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
    if (ctx->p_B <= 0 || !ctx->prop_batch)
        return ctx->fail(LVBGPU_E_STATE, "no device batch: call lvbgpu_propose_score first");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::vector<ProposalInfo> info((size_t)ctx->p_B);
    HIPCHK(ctx, hipMemcpyAsync(info.data(), ctx->d_pinfo.p, info.size() * sizeof(ProposalInfo), hipMemcpyDeviceToHost,
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
