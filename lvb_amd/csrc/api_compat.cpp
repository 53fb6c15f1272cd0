// api_compat.cpp - liblvbgpu.so: strict compatibility, the reference's own tree block in and out (lvbgpu_getplen_compat).
#include "ctx.hpp"

// =================================================================================== strict compat

extern "C" int lvbgpu_getplen_compat(lvbgpu_ctx *ctx, void *tree_v, long root, int64_t *length_out)
{
    if (!ctx || !tree_v || !length_out)
        return LVBGPU_E_ARG;
    ENTER(ctx);
    RefNode *tree = (RefNode *)tree_v;
    const int32_t nb = ctx->nb, n = (int32_t)ctx->n;
    const uint32_t W = (uint32_t)ctx->nwords, Wp = ctx->stride_words;

    // topology + dirty flags + cached changes of clean nodes (TreeEvaluation.c:191-202)
    std::vector<int32_t> l(nb), r(nb);
    std::vector<uint8_t> dirty(nb, 0);
    long long base = 0;
    for (int32_t i = 0; i < nb; i++)
    {
        l[i] = (int32_t)tree[i].left;
        r[i] = (int32_t)tree[i].right;
        if (i >= n)
        {
            if (tree[i].sitestate[0] == 0)
                dirty[i] = 1;
            else
                base += tree[i].changes;
        }
    }
    std::string why;
    Topology t;
    if (!t.assign(n, l.data(), r.data(), (int32_t)root, &why))
        return ctx->fail(LVBGPU_E_TOPOLOGY, why);
    Program prog;
    ctx->pb.build_flagged(t, dirty.data(), prog);
    int rc = check_depth(ctx, prog.max_stack);
    if (rc != LVBGPU_OK)
        return rc;

    // operand rows -> input slots, produced nodes -> output slots
    if ((int32_t)ctx->slot_of.size() != nb)
    {
        ctx->slot_of.assign(nb, 0);
        ctx->slot_epoch.assign(nb, 0);
        ctx->slot_gen = 0;
    }
    if (++ctx->slot_gen == 0)
    {
        std::fill(ctx->slot_epoch.begin(), ctx->slot_epoch.end(), 0u);
        ctx->slot_gen = 1;
    }
    std::vector<int32_t> in_nodes;
    for (uint32_t &tk : prog.toks)
    {
        const int32_t node = (int32_t)(tk & TOK_ROW_MASK);
        if (ctx->slot_epoch[node] != ctx->slot_gen)
        {
            ctx->slot_epoch[node] = ctx->slot_gen;
            ctx->slot_of[node] = (int32_t)in_nodes.size();
            in_nodes.push_back(node);
        }
        tk = (tk & ~TOK_ROW_MASK) | (uint32_t)ctx->slot_of[node];
    }
    std::vector<int32_t> out_nodes;
    for (int32_t &d : prog.dsts)
        if (d >= 0)
        {
            out_nodes.push_back(d);
            d = (int32_t)out_nodes.size() - 1;
        }
    const uint32_t n_in = (uint32_t)in_nodes.size(), n_out = (uint32_t)out_nodes.size();

    // one input arena [cand][toks][dsts][rows], one output arena [len][changes x (n_out+1)][rows]
    const size_t o_t = align16(sizeof(CandDesc));
    const size_t o_d = o_t + align16(prog.toks.size() * 4);
    const size_t o_rows = o_d + align16(prog.dsts.size() * 4);
    const size_t in_bytes = o_rows + (size_t)n_in * Wp * 8;
    const size_t oo_ch = 16;
    const size_t oo_rows = align16(oo_ch + (size_t)(n_out + 1) * 8);
    const size_t out_bytes = oo_rows + (size_t)n_out * Wp * 8;
    HIPCHK(ctx, ctx->h_cin.reserve(in_bytes));
    HIPCHK(ctx, ctx->d_cin.reserve(in_bytes));
    HIPCHK(ctx, ctx->h_cout.reserve(out_bytes));
    HIPCHK(ctx, ctx->d_cout.reserve(out_bytes));

    char *hin = (char *)ctx->h_cin.p;
    CandDesc cd{};
    cd.tok_off = 0;
    cd.ntok = (uint32_t)prog.toks.size();
    cd.dst_off = 0;
    cd.ncomb = (uint32_t)prog.dsts.size();
    cd.base = base;
    cd.flags = 0;
    for (uint32_t tk : prog.toks)
        cd.nfresh += (tk & TOK_FRESH) ? 1u : 0u;
    memcpy(hin, &cd, sizeof cd);
    memcpy(hin + o_t, prog.toks.data(), prog.toks.size() * 4);
    memcpy(hin + o_d, prog.dsts.data(), prog.dsts.size() * 4);
    for (uint32_t s = 0; s < n_in; s++)
    {
        uint64_t *dst = (uint64_t *)(hin + o_rows) + (size_t)s * Wp;
        memcpy(dst, tree[in_nodes[s]].sitestate, (size_t)W * 8);
        for (uint32_t w = W; w < Wp; w++)
            dst[w] = ~0ull;
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_cin.p, hin, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_cout.p, 0, oo_rows, ctx->stream));
    // operands arrive in the reference's nibble layout; the walk works on bit planes
    HIPCHK(ctx, launch_relayout((uint4 *)((char *)ctx->d_cin.p + o_rows), n_in, Wp / 2, true, ctx->stream));

    WalkArgs a{};
    a.rows_in = (const uint4 *)((const char *)ctx->d_cin.p + o_rows);
    a.rows_out = (uint4 *)((char *)ctx->d_cout.p + oo_rows);
    a.cands = (const CandDesc *)ctx->d_cin.p;
    a.toks = (const uint32_t *)((const char *)ctx->d_cin.p + o_t);
    a.dsts = (const int32_t *)((const char *)ctx->d_cin.p + o_d);
    a.node_changes = nullptr;
    a.s_all = nullptr;
    a.len_out = (unsigned long long *)ctx->d_cout.p;
    a.changes_out = (unsigned long long *)((char *)ctx->d_cout.p + oo_ch);
    a.root_slot = n_out;
    a.n_first = UINT32_MAX; // one program block
    a.in_stride4 = Wp / 2;                // the staging arenas are row-major
    a.in_tile_bytes = 1024;
    a.block_bytes = (uint64_t)n_in * Wp * 8u;
    a.nrows = n_in; // rows of this call's staging block
    a.bias_from = UINT32_MAX; // tokens name staging slots, not nodes of a resident tree
    a.chain_rows = 0;
    a.out_stride4 = Wp / 2;
    a.out_tile4 = 64;
    a.B = 1;
    a.ntiles = ctx->ntiles;
    a.ngroups = ctx->ntiles;
    a.nitems = ctx->ntiles;
    a.stack_depth = (uint32_t)std::max(prog.max_stack, 1);
    HIPCHK(ctx, launch_walk(a, true, ctx->stream));
    HIPCHK(ctx, launch_relayout((uint4 *)((char *)ctx->d_cout.p + oo_rows), n_out, Wp / 2, false, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_cout.p, ctx->d_cout.p, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, wait_for_step(ctx, 1)); // one small step: poll for it

    // write back exactly what the reference's getplen leaves behind (TreeEvaluation.c:228-229)
    const char *hout = (const char *)ctx->h_cout.p;
    const long long total = *(const long long *)hout;
    const unsigned long long *och = (const unsigned long long *)(hout + oo_ch);
    for (uint32_t s = 0; s < n_out; s++)
    {
        const int32_t node = out_nodes[s];
        memcpy(tree[node].sitestate, (const uint64_t *)(hout + oo_rows) + (size_t)s * Wp, (size_t)W * 8);
        tree[node].changes = (long)och[s];
    }
    *length_out = total;
    if (total <= 0)
        return ctx->fail(LVBGPU_E_ZEROLEN, "assertion failed: changes > 0");
    return LVBGPU_OK;
}

// =================================================================================== timing

