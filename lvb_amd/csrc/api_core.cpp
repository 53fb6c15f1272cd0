// api_core.cpp - liblvbgpu.so: errors, encoding, context life cycle, the resident tree (set, read back, commit), timers.
//
// There is deliberately no CPU implementation of any scoring entry point in this library: without
// a working HIP device every one of them fails with LVBGPU_E_NODEVICE / LVBGPU_E_HIP.
#include "ctx.hpp"

namespace lvbgpu_detail
{

thread_local std::string g_last_error_noctx;

int hip_status_noctx(hipError_t e, const char *what)
{
    g_last_error_noctx = std::string(what) + ": " + hipGetErrorString(e);
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver)
               ? LVBGPU_E_NODEVICE
               : (e == hipErrorOutOfMemory ? LVBGPU_E_NOMEM : LVBGPU_E_HIP);
}

// end of a search step on the context's stream
// Small steps are polled for (the runtime's wake-up costs ~10 us), big ones sleep between polls; either way the wait
// ends at the context's limit with hipErrorNotReady, which fail_hip words as what it is.
hipError_t wait_for_step(lvbgpu_ctx *ctx, int32_t B)
{
    const WaitClock clock(ctx->wait_limit_s);
    hipError_t q;
    uint32_t spins = 0;
    while ((q = hipStreamQuery(ctx->stream)) == hipErrorNotReady)
    {
        if (B > SPIN_WAIT_MAX_B)
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        if ((++spins & 1023u) == 0 && clock.expired())
        {
            ctx->wait_gave_up = true;
            break;
        }
    }
    return q;
}

WalkArgs resident_args(lvbgpu_ctx *ctx, const void *prog, size_t off_toks, size_t off_dsts, void *d_len,
                       uint32_t B, int32_t max_stack)
{
    WalkArgs a{};
    a.rows_in = (const uint4 *)ctx->d_rows;
    a.rows_out = (uint4 *)ctx->d_rows;
    a.cands = (const CandDesc *)prog;
    a.toks = (const uint32_t *)((const char *)prog + off_toks);
    a.dsts = (const int32_t *)((const char *)prog + off_dsts);
    a.node_changes = (const long long *)ctx->d_changes;
    a.s_all = ctx->d_scalars;
    a.len_out = (unsigned long long *)d_len;
    a.changes_out = ctx->d_changes;
    a.in_stride4 = 64;                                   // tile-major: [tile][row][64 groups of 16 bytes]
    a.in_tile_bytes = (uint64_t)ctx->rows_total() * 1024u;
    a.block_bytes = (uint64_t)ctx->rows_total() * ctx->stride_words * 8u;
    a.nrows = ctx->rows_total();
    a.bias_from = (uint32_t)ctx->n;
    a.chain_rows = ctx->chain_rows();
    a.out_stride4 = 64;
    a.out_tile4 = (uint64_t)ctx->rows_total() * 64u;
    a.B = B;
    a.ntiles = ctx->ntiles;
    a.ngroups = choose_groups(B, ctx->ntiles, ctx->target_waves);
    a.nitems = B * a.ngroups;
    a.stack_depth = (uint32_t)std::max(max_stack, 1);
    a.root_slot = ctx->rows_total(); // + the candidate's chain
    a.n_first = UINT32_MAX;          // one program block
    return a;
}

int check_depth(lvbgpu_ctx *ctx, int32_t max_stack)
{
    // two more KiB per wave than the stack itself: a commit walk parks at least one produced set (and the counts)
    if ((size_t)(std::max(max_stack, 1) + 2) * WALK_WAVES * 64 * sizeof(uint4) > MAX_LDS_BYTES)
        return ctx->fail(LVBGPU_E_ARG, "postorder program needs a deeper operand stack than LDS holds");
    return LVBGPU_OK;
}

int context_common_init(lvbgpu_ctx *ctx, int device, long n, long nwords)
{
    if (n < 3 || nwords < 1 || 2 * n - 3 > MAX_ROWS)
        return ctx->fail(LVBGPU_E_ARG, "n or nwords out of range");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return ctx->fail(LVBGPU_E_NODEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
    if (device < 0 || device >= count)
        return ctx->fail(LVBGPU_E_ARG, "device index out of range");
    ctx->device = device;
    HIPCHK(ctx, hipSetDevice(device));
    ctx->n = n;
    ctx->nwords = nwords;
    ctx->nb = (int32_t)(2 * n - 3);
    ctx->ntiles = round_up((uint32_t)nwords, TILE_WORDS) / TILE_WORDS;
    // a row is ntiles whole tiles long (padding holds all-ones); tuning knob for experiments: the wave-count target
    ctx->stride_words = ctx->ntiles * TILE_WORDS;
    ctx->stride4 = ctx->stride_words / 2;
    if (const char *tw = getenv("LVBGPU_TARGET_WAVES"))
        ctx->target_waves = (uint32_t)std::max(1, atoi(tw));
    if (const char *wl = getenv("LVBGPU_WAIT_SECONDS"))
        if (atof(wl) > 0.0)
            ctx->wait_limit_s = atof(wl);
    // (row offsets are 64-bit in the kernels: the tree block is limited by HBM, not by index width)
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    HIPCHK(ctx, hipEventCreate(&ctx->ev0));
    HIPCHK(ctx, hipEventCreate(&ctx->ev1));
    ctx->nchains = 1;
    ctx->chain = 0;
    ctx->parked.assign(1, ChainSlot{});
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_rows, (size_t)ctx->rows_total() * ctx->stride_words * 8));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_changes, (size_t)(ctx->rows_total() + ctx->nchains) * 8));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_scalars, 4 * 8)); // [2]: finished-wave count of direct steps
    HIPCHK(ctx, hipMemsetAsync(ctx->d_changes, 0, (size_t)(ctx->rows_total() + ctx->nchains) * 8, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_scalars, 0, 32, ctx->stream));
    HIPCHK(ctx, ctx->h_step.reserve(64));
    memset(ctx->h_step.p, 0, 64);
    if (const char *ds = getenv("LVBGPU_DIRECT_STEPS"))
        ctx->direct_steps = ds[0] != '0';
    ctx->starve_watcher = getenv("LVBGPU_DEBUG_STARVE_WATCHER") != nullptr;
    if (const char *lp = getenv("LVBGPU_LPT"))
        ctx->lpt_order = lp[0] != '0';
    if (const char *pr = getenv("LVBGPU_PAIR")) // n: batches of n candidates and more are walked two candidates per wave (unset / 0: none)
        ctx->pair_min = std::max(0, atoi(pr));
    if (const char *pl = getenv("LVBGPU_PIPELINE"))
        ctx->pipeline_steps = pl[0] != '0';
    static_assert(lvbgpu_ctx::STEP_PIPELINE == 4, "lvbgpu_destroy lists the step batches");
    HIPCHK(ctx, upload_iupac_table());
    HIPCHK(ctx, raise_lds_limit());
    ctx->pb.resize(ctx->nb);
    return LVBGPU_OK;
}

// The leaf rows arrive row-major in `leaves` ([n][stride_words], the reference's nibble layout).  Everything that is not a
// leaf word becomes all-ones (inert under fitch: padding words here, the internal rows with the memset of the whole
// resident block); the leaf rows go from nibbles to bit planes (all-ones stays all-ones) and from there to their places
// in the tile-major resident block.
int finish_rows(lvbgpu_ctx *ctx, DevBuf &leaves)
{
    HIPCHK(ctx, launch_fill_pad((uint64_t *)leaves.p, (uint32_t)ctx->n, (uint32_t)ctx->nwords, ctx->stride_words, (uint32_t)ctx->n,
                                ctx->stream));
    HIPCHK(ctx, launch_relayout((uint4 *)leaves.p, (uint32_t)ctx->n, ctx->stride4, true, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_rows, 0xFF, (size_t)ctx->rows_total() * ctx->stride_words * 8, ctx->stream));
    HIPCHK(ctx, launch_rows_to_tiles((const uint4 *)leaves.p, (uint4 *)ctx->d_rows, (uint32_t)ctx->n, ctx->rows_total(), ctx->ntiles,
                                     ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    leaves.release();
    return LVBGPU_OK;
}

} // namespace lvbgpu_detail

// =================================================================================== library

extern "C" const char *lvbgpu_strerror(int status)
{
    switch (status)
    {
    case LVBGPU_OK: return "ok";
    case LVBGPU_E_ARG: return "bad argument";
    case LVBGPU_E_NODEVICE: return "no usable HIP device";
    case LVBGPU_E_HIP: return "HIP call failed";
    case LVBGPU_E_NOMEM: return "out of memory";
    case LVBGPU_E_STATE: return "call order violated";
    case LVBGPU_E_TOPOLOGY: return "not a binary tree rooted at a leaf";
    case LVBGPU_E_SYMBOL: return "bad base symbol in data matrix";
    case LVBGPU_E_ZEROLEN: return "tree length is not positive";
    case LVBGPU_E_COMM: return "RCCL failure";
    default: return "unknown status";
    }
}

extern "C" const char *lvbgpu_last_error(const lvbgpu_ctx *ctx)
{
    return ctx ? ctx->last_error.c_str() : g_last_error_noctx.c_str();
}

extern "C" int lvbgpu_device_count(void)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess)
        return hip_status_noctx(e, "hipGetDeviceCount");
    return count;
}

extern "C" int lvbgpu_abi_version(void) { return ABI_VERSION; }

extern "C" long lvbgpu_words_per_row(long m)
{
    // reference DataOperations.c:446-458: m/16 rounded up
    return (m >> 4) + ((m & 15) ? 1 : 0);
}

// =================================================================================== encoding

namespace lvbgpu_detail
{
int encode_into(lvbgpu_ctx *ctx, long n, long m, const char *const *rows, uint64_t *d_rows, uint32_t stride_words)
{
    const long nwords = lvbgpu_words_per_row(m);
    DevBuf d_text, d_bad;
    PinBuf h_text;
    int rc = LVBGPU_OK;
    const size_t tbytes = (size_t)n * (size_t)m;
    hipError_t e;
    if ((e = h_text.reserve(tbytes)) != hipSuccess || (e = d_text.reserve(tbytes)) != hipSuccess ||
        (e = d_bad.reserve(8)) != hipSuccess)
        rc = ctx->fail_hip(e, "encode staging");
    unsigned long long bad = ~0ull;
    if (rc == LVBGPU_OK)
    {
        for (long i = 0; i < n; i++)
            memcpy((char *)h_text.p + (size_t)i * m, rows[i], (size_t)m);
        if ((e = hipMemcpyAsync(d_text.p, h_text.p, tbytes, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess ||
            (e = hipMemcpyAsync(d_bad.p, &bad, 8, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess ||
            (e = launch_encode_text((const uint8_t *)d_text.p, (uint32_t)n, (uint64_t)m, (uint32_t)nwords,
                                    stride_words, d_rows, (unsigned long long *)d_bad.p, ctx->stream)) !=
                hipSuccess ||
            (e = hipMemcpyAsync(&bad, d_bad.p, 8, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess ||
            (e = hipStreamSynchronize(ctx->stream)) != hipSuccess)
            rc = ctx->fail_hip(e, "encode_text");
    }
    if (rc == LVBGPU_OK && bad != ~0ull)
    {
        const unsigned long long pos = bad - 1;
        char msg[160];
        snprintf(msg, sizeof msg, "bad base symbol in data MSA: '%c' (row %llu, column %llu)",
                 rows[pos / m][pos % m], pos / m, pos % m);
        rc = ctx->fail(LVBGPU_E_SYMBOL, msg);
    }
    d_text.release();
    d_bad.release();
    h_text.release();
    return rc;
}
} // namespace lvbgpu_detail

extern "C" int lvbgpu_encode_text(int device, long n, long m, const char *const *rows, uint64_t *out)
{
    if (!rows || !out || n < 1 || m < 1)
        return LVBGPU_E_ARG;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return hip_status_noctx(e == hipSuccess ? hipErrorNoDevice : e, "hipGetDeviceCount");
    if (device < 0 || device >= count)
        return LVBGPU_E_ARG;
    lvbgpu_ctx tmp;
    tmp.device = device;
    int rc = LVBGPU_OK;
    const long nwords = lvbgpu_words_per_row(m);
    DevBuf d_out;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreate(&tmp.stream)) != hipSuccess ||
        (e = upload_iupac_table()) != hipSuccess || (e = d_out.reserve((size_t)n * nwords * 8)) != hipSuccess)
        rc = tmp.fail_hip(e, "encode setup");
    if (rc == LVBGPU_OK)
        rc = encode_into(&tmp, n, m, rows, (uint64_t *)d_out.p, (uint32_t)nwords);
    if (rc == LVBGPU_OK &&
        (e = hipMemcpy(out, d_out.p, (size_t)n * nwords * 8, hipMemcpyDeviceToHost)) != hipSuccess)
        rc = tmp.fail_hip(e, "encode download");
    g_last_error_noctx = tmp.last_error;
    d_out.release();
    if (tmp.stream)
        (void)hipStreamDestroy(tmp.stream);
    return rc;
}

// =================================================================================== context

extern "C" void lvbgpu_destroy(lvbgpu_ctx *ctx)
{
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->device);
    if (ctx->comm)
        (void)lvbgpu_comm_destroy(ctx);
    if (ctx->stream)
        (void)hipStreamSynchronize(ctx->stream);
    if (ctx->copy_stream)
        (void)hipStreamSynchronize(ctx->copy_stream);

    if (ctx->d_rows)
        (void)hipFree(ctx->d_rows);
    if (ctx->d_changes)
        (void)hipFree(ctx->d_changes);
    if (ctx->d_scalars)
        (void)hipFree(ctx->d_scalars);
    delete ctx->pool;
    ctx->pool = nullptr;
    ctx->d_topo4.release();
    ctx->d_table_ready.release();
    for (lvbgpu_ctx::PropSlot &ps : ctx->pslot)
    {
        ps.d_pedits.release();
        ps.d_pinfo.release();
        if (ps.done_ev)
            (void)hipEventDestroy(ps.done_ev);
        if (ps.walk_ev)
            (void)hipEventDestroy(ps.walk_ev);
        ps.h_flag.release();
    }
    ctx->h_pinfo.release();
    ctx->h_topo.release();
    ctx->d_gen_prof.release();
    ctx->d_post_prof.release();
    ctx->d_done.release();
    for (int i = 0; i < lvbgpu_ctx::PICK_SLOTS; i++)
        ctx->h_pick[i].release();
    ctx->d_moves.release();
    ctx->h_moves.release();
    ctx->h_step.release();
    ctx->d_tmp_changes.release();
    for (lvbgpu_batch *rb : {ctx->step_batch[0], ctx->step_batch[1], ctx->step_batch[2], ctx->step_batch[3], ctx->full_batch,
                             ctx->pslot[0].batch, ctx->pslot[1].batch})
        if (rb)
        {
            rb->ctx = nullptr;
            lvbgpu_batch_free(rb);
        }
    ctx->d_len.release();
    ctx->d_export.release();
    for (int i = 0; i < lvbgpu_ctx::COMMIT_SLOTS; i++)
    {
        ctx->h_commit[i].release();
        ctx->d_commit[i].release();
        if (ctx->commit_ev[i])
            (void)hipEventDestroy(ctx->commit_ev[i]);
    }
    ctx->h_pin.release();
    ctx->d_cin.release();
    ctx->d_cout.release();
    ctx->h_cin.release();
    ctx->h_cout.release();
    ctx->d_comm.release();
    ctx->d_probe_sink.release();
    for (lvbgpu_batch *zb : ctx->set_aside) // step batches abandoned after a wait that gave up (the streams have been waited for above)
    {
        zb->ctx = nullptr;
        lvbgpu_batch_free(zb);
    }
    ctx->set_aside.clear();
    for (lvbgpu_batch *hb : ctx->held) // the caller still owns them; they must not reach into a dead context
        hb->ctx = nullptr;
    ctx->held.clear();
    for (hipEvent_t ev : ctx->wt_ev)
        if (ev)
            (void)hipEventDestroy(ev);
    if (ctx->ev0)
        (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1)
        (void)hipEventDestroy(ctx->ev1);
    if (ctx->copy_stream)
        (void)hipStreamDestroy(ctx->copy_stream);

    if (ctx->stream)
        (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int lvbgpu_create(lvbgpu_ctx **out, int device, long n, long nwords, const uint64_t *leaf_matrix,
                             long row_stride_words)
{
    if (!out || !leaf_matrix || row_stride_words < nwords)
        return LVBGPU_E_ARG;
    *out = nullptr;
    lvbgpu_ctx *ctx = new (std::nothrow) lvbgpu_ctx();
    if (!ctx)
        return LVBGPU_E_NOMEM;
    int rc = context_common_init(ctx, device, n, nwords);
    DevBuf leaves; // row-major staging of the leaf rows (released by finish_rows)
    if (rc == LVBGPU_OK)
    {
        hipError_t e = leaves.reserve((size_t)n * ctx->stride_words * 8);
        if (e == hipSuccess)
            e = hipMemcpy2DAsync(leaves.p, (size_t)ctx->stride_words * 8, leaf_matrix, (size_t)row_stride_words * 8,
                                 (size_t)nwords * 8, (size_t)n, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess)
            rc = ctx->fail_hip(e, "upload leaf matrix");
    }
    if (rc == LVBGPU_OK)
        rc = finish_rows(ctx, leaves);
    leaves.release();
    if (rc != LVBGPU_OK)
    {
        g_last_error_noctx = ctx->last_error;
        lvbgpu_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return LVBGPU_OK;
}

extern "C" int lvbgpu_create_from_text(lvbgpu_ctx **out, int device, long n, long m, const char *const *rows)
{
    if (!out || !rows || m < 1)
        return LVBGPU_E_ARG;
    *out = nullptr;
    lvbgpu_ctx *ctx = new (std::nothrow) lvbgpu_ctx();
    if (!ctx)
        return LVBGPU_E_NOMEM;
    int rc = context_common_init(ctx, device, n, lvbgpu_words_per_row(m));
    DevBuf leaves; // row-major staging of the leaf rows (released by finish_rows)
    if (rc == LVBGPU_OK)
    {
        const hipError_t e = leaves.reserve((size_t)n * ctx->stride_words * 8);
        if (e != hipSuccess)
            rc = ctx->fail_hip(e, "leaf row staging");
    }
    if (rc == LVBGPU_OK)
        rc = encode_into(ctx, n, m, rows, (uint64_t *)leaves.p, ctx->stride_words);
    if (rc == LVBGPU_OK)
        rc = finish_rows(ctx, leaves);
    leaves.release();
    if (rc != LVBGPU_OK)
    {
        g_last_error_noctx = ctx->last_error;
        lvbgpu_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return LVBGPU_OK;
}

// A second context on the same alignment: the leaf rows are copied on the device (a tile's leaf rows are one piece of
// the tile-major block), everything else starts empty - one resident-tree slot, no tree.
extern "C" int lvbgpu_fork(lvbgpu_ctx *src, lvbgpu_ctx **out)
{
    if (!src || !out)
        return LVBGPU_E_ARG;
    *out = nullptr;
    HIPCHK(src, hipSetDevice(src->device));
    lvbgpu_ctx *ctx = new (std::nothrow) lvbgpu_ctx();
    if (!ctx)
        return LVBGPU_E_NOMEM;
    int rc = context_common_init(ctx, src->device, src->n, src->nwords);
    if (rc == LVBGPU_OK)
    {
        ctx->target_waves = src->target_waves;
        ctx->wait_limit_s = src->wait_limit_s;
        hipError_t e = hipMemsetAsync(ctx->d_rows, 0xFF, (size_t)ctx->rows_total() * ctx->stride_words * 8, ctx->stream);
        if (e == hipSuccess)
            e = hipMemcpy2DAsync(ctx->d_rows, (size_t)ctx->rows_total() * 1024u, src->d_rows, (size_t)src->rows_total() * 1024u,
                                 (size_t)src->n * 1024u, (size_t)src->ntiles, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess)
            rc = ctx->fail_hip(e, "copy of the leaf rows");
    }
    if (rc != LVBGPU_OK)
    {
        src->last_error = ctx->last_error;
        lvbgpu_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return LVBGPU_OK;
}

extern "C" long lvbgpu_n(const lvbgpu_ctx *ctx) { return ctx ? ctx->n : 0; }
extern "C" long lvbgpu_nwords(const lvbgpu_ctx *ctx) { return ctx ? ctx->nwords : 0; }

// =================================================================================== resident tree

namespace lvbgpu_detail
{
// run one stored-result program (full evaluation or commit) against the resident rows and
// refresh S_all / current length.  `prog` holds node ids.
// refresh cur_length from the device scalars (after an asynchronous commit)
int read_current_length(lvbgpu_ctx *ctx)
{
    if (!ctx->cur_length_stale)
        return LVBGPU_OK;
    // length = S_all (kept current by every commit) + the root slot of changes[]
    long long s_all = 0, root_changes = 0;
    HIPCHK(ctx, hipMemcpyAsync(&s_all, ctx->scalars(), 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&root_changes, ctx->d_changes + ctx->root_slot(), 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->cur_length = s_all + root_changes;
    ctx->cur_length_stale = false;
    return LVBGPU_OK;
}

// readback = false: everything is only enqueued (own pinned slot for the program, no
// synchronisation); the caller already knows the length from scoring the candidate
int run_commit_program(lvbgpu_ctx *ctx, const Program &prog, bool zero_all, bool readback)
{
    int rc = check_depth(ctx, prog.max_stack);
    if (rc != LVBGPU_OK)
        return rc;
    Packed pk;
    pk.add(prog, 0, 0, 0, (uint32_t)ctx->chain << CAND_CHAIN_SHIFT);
    // the program goes through one of a few pinned slots, each guarded by an event, so the host
    // never waits for the device here
    const size_t o_t = align16(sizeof(CandDesc));
    const size_t o_d = o_t + align16(prog.toks.size() * 4);
    const size_t total = o_d + align16(prog.dsts.size() * 4);
    const int slot = ctx->commit_slot;
    ctx->commit_slot = (slot + 1) % lvbgpu_ctx::COMMIT_SLOTS;
    if (!ctx->commit_ev[slot])
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->commit_ev[slot], hipEventDisableTiming));
    else
        HIPCHK(ctx, hipEventSynchronize(ctx->commit_ev[slot])); // long done unless 4 commits are in flight
    HIPCHK(ctx, ctx->h_commit[slot].reserve(total));
    HIPCHK(ctx, ctx->d_commit[slot].reserve(total));
    char *h = (char *)ctx->h_commit[slot].p;
    memcpy(h, pk.cands.data(), sizeof(CandDesc));
    memcpy(h + o_t, prog.toks.data(), prog.toks.size() * 4);
    memcpy(h + o_d, prog.dsts.data(), prog.dsts.size() * 4);
    DevBuf &dprog = ctx->d_commit[slot];
    HIPCHK(ctx, ctx->d_len.reserve(8));
    if (zero_all)
    {
        // full evaluation: everything is cleared, every node recomputed, the counts summed once
        HIPCHK(ctx, hipMemcpyAsync(dprog.p, h, total, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipEventRecord(ctx->commit_ev[slot], ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(ctx->d_len.p, 0, 8, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(ctx->d_changes + ctx->row_of((int32_t)ctx->n), 0, (size_t)ctx->chain_rows() * 8, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(ctx->d_changes + ctx->root_slot(), 0, 8, ctx->stream));
        WalkArgs a = resident_args(ctx, dprog.p, o_t, o_d, ctx->d_len.p, 1, prog.max_stack);
        HIPCHK(ctx, launch_walk(a, true, ctx->stream));
        HIPCHK(ctx, launch_sum_changes(ctx->d_changes, ctx->row_of((int32_t)ctx->n), ctx->row_of((int32_t)ctx->n) + ctx->chain_rows(),
                                       ctx->root_slot(), ctx->scalars(), ctx->stream));
    }
    else if (!ctx->direct_steps)
    {
        // LVBGPU_DIRECT_STEPS=0: copy, one small launch that clears what the walk accumulates into and takes
        // the recomputed nodes' old counts out of S_all, then the walk
        HIPCHK(ctx, hipMemcpyAsync(dprog.p, h, total, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipEventRecord(ctx->commit_ev[slot], ctx->stream));
        HIPCHK(ctx, launch_zero_changes((unsigned long long *)ctx->d_changes, (const int32_t *)((const char *)dprog.p + o_d),
                                        (uint32_t)prog.dsts.size(), (unsigned long long *)ctx->d_changes + ctx->root_slot(),
                                        (unsigned long long *)ctx->d_len.p, (unsigned long long *)ctx->scalars(),
                                        (uint32_t)ctx->n, (uint32_t)ctx->chain * ctx->chain_rows(), ctx->stream));
        WalkArgs a = resident_args(ctx, dprog.p, o_t, o_d, ctx->d_len.p, 1, prog.max_stack);
        a.s_all_out = (unsigned long long *)ctx->d_scalars; // the walk adds the new counts (to its chain's slot): S_all stays current
        HIPCHK(ctx, launch_walk(a, true, ctx->stream));
    }
    else
    {
        // the accept path is ONE launch: the walk's last wave settles changes[] and S_all itself (fused
        // commit, kernels.hpp), and a small program is read where it lies in the pinned slot
        const uint32_t ngroups = choose_groups(1, ctx->ntiles, ctx->target_waves);
        const bool in_place = total * ngroups <= DIRECT_READ_MAX_BYTES;
        if (!in_place)
            HIPCHK(ctx, hipMemcpyAsync(dprog.p, h, total, hipMemcpyHostToDevice, ctx->stream));
        // one accumulator per combine of the program (<= nb + 1); "cleared" is a property of the allocation, so it
        // is the capacity that is remembered, not a flag that could outlive a buffer that grew
        HIPCHK(ctx, ctx->d_tmp_changes.reserve(std::max((size_t)(ctx->nb + 1), prog.dsts.size()) * 8));
        if (ctx->tmp_changes_zeroed_cap != ctx->d_tmp_changes.cap)
        {
            HIPCHK(ctx, hipMemsetAsync(ctx->d_tmp_changes.p, 0, ctx->d_tmp_changes.cap, ctx->stream));
            ctx->tmp_changes_zeroed_cap = ctx->d_tmp_changes.cap;
        }
        WalkArgs a = resident_args(ctx, in_place ? (const void *)h : dprog.p, o_t, o_d, ctx->d_len.p, 1, prog.max_stack);
        a.s_all_out = (unsigned long long *)ctx->d_scalars;
        a.tmp_changes = (unsigned long long *)ctx->d_tmp_changes.p;
        a.done_count = (uint32_t *)(ctx->d_scalars + 2);
        HIPCHK(ctx, launch_walk(a, true, ctx->stream));
        HIPCHK(ctx, hipEventRecord(ctx->commit_ev[slot], ctx->stream)); // the slot is free once the walk has read it
    }
    ctx->cur_length_stale = true;
    return readback ? read_current_length(ctx) : LVBGPU_OK;
}
} // namespace lvbgpu_detail

namespace lvbgpu_detail
{
// the selected chain's state lives in the context's own fields: put it away / fetch another
void park_chain(lvbgpu_ctx *ctx)
{
    ChainSlot &s = ctx->parked[(size_t)ctx->chain];
    s.topo = std::move(ctx->topo);
    s.topo_version = ctx->topo_version;
    s.have_tree = ctx->have_tree;
    s.cur_length = ctx->cur_length;
    s.cur_length_stale = ctx->cur_length_stale;
    s.d_topo_version = ctx->d_topo_version;
    s.gen_table_bytes = ctx->gen_table_bytes;
    s.gen_K = ctx->gen_K;
}
void unpark_chain(lvbgpu_ctx *ctx, int32_t c)
{
    ChainSlot &s = ctx->parked[(size_t)c];
    ctx->chain = c;
    ctx->topo = std::move(s.topo);
    ctx->topo_version = s.topo_version;
    ctx->have_tree = s.have_tree;
    ctx->cur_length = s.cur_length;
    ctx->cur_length_stale = s.cur_length_stale;
    ctx->d_topo_version = s.d_topo_version;
    ctx->gen_table_bytes = s.gen_table_bytes;
    ctx->gen_K = s.gen_K;
}
} // namespace lvbgpu_detail

extern "C" int lvbgpu_set_chains(lvbgpu_ctx *ctx, int32_t nchains)
{
    if (!ctx || nchains < 1 || nchains > MAX_CHAINS)
        return LVBGPU_E_ARG;
    {
        const int rf = settle(ctx); // the host side of a chain commit still on its way
        if (rf != LVBGPU_OK)
            return rf;
    }
    ENTER(ctx);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));
    if ((uint64_t)(ctx->n + (long)nchains * (ctx->n - 3)) >= (uint64_t)MAX_ROWS)
        return ctx->fail(LVBGPU_E_ARG, "too many rows for the token format");
    // new blocks: leaf rows move over, every tree slot starts empty (all-ones rows, no resident tree)
    const int32_t old_chains = ctx->nchains;
    uint64_t *rows = nullptr;
    unsigned long long *changes = nullptr;
    long long *scalars = nullptr;
    ctx->nchains = nchains;
    const size_t row_bytes = (size_t)ctx->stride_words * 8;
    hipError_t e = hipMalloc((void **)&rows, (size_t)ctx->rows_total() * row_bytes);
    if (e == hipSuccess)
        e = hipMalloc((void **)&changes, (size_t)(ctx->rows_total() + nchains) * 8);
    if (e == hipSuccess)
        e = hipMalloc((void **)&scalars, (size_t)nchains * 32);
    if (e != hipSuccess)
    {
        ctx->nchains = old_chains;
        (void)hipFree(rows);
        (void)hipFree(changes);
        (void)hipFree(scalars);
        return ctx->fail_hip(e, "lvbgpu_set_chains: allocate");
    }
    // tile-major blocks: all-ones everywhere, then every tile's n leaf slices (1 KiB each, contiguous) move to the front
    // of that tile in the new block
    HIPCHK(ctx, hipMemsetAsync(rows, 0xFF, (size_t)ctx->rows_total() * row_bytes, ctx->stream));
    {
        const size_t old_total = (size_t)ctx->n + (size_t)old_chains * (size_t)(ctx->n - 3);
        HIPCHK(ctx, hipMemcpy2DAsync(rows, (size_t)ctx->rows_total() * 1024u, ctx->d_rows, old_total * 1024u, (size_t)ctx->n * 1024u,
                                     (size_t)ctx->ntiles, hipMemcpyDeviceToDevice, ctx->stream));
    }
    HIPCHK(ctx, hipMemsetAsync(changes, 0, (size_t)(ctx->rows_total() + nchains) * 8, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(scalars, 0, (size_t)nchains * 32, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(ctx->d_rows);
    (void)hipFree(ctx->d_changes);
    (void)hipFree(ctx->d_scalars);
    ctx->d_rows = rows;
    ctx->d_changes = changes;
    ctx->d_scalars = scalars;
    ctx->parked.assign((size_t)nchains, ChainSlot{});
    ctx->chain = 0;
    ctx->topo = Topology{};
    ctx->topo_version = ++ctx->version_counter;
    ctx->have_tree = false;
    ctx->cur_length = 0;
    ctx->cur_length_stale = false;
    ctx->d_topo_version = ~0ull;
    for (lvbgpu_ctx::PropSlot &ps : ctx->pslot)
    {
        ps.p_B = 0;
        ps.segs.clear();
        ps.in_flight = false;
    }
    ctx->tmp_changes_zeroed_cap = 0;
    return LVBGPU_OK;
}

extern "C" int lvbgpu_select_chain(lvbgpu_ctx *ctx, int32_t chain)
{
    if (!ctx || chain < 0 || chain >= ctx->nchains)
        return LVBGPU_E_ARG;
    {
        const int rf = settle(ctx); // the host side of a chain commit still on its way
        if (rf != LVBGPU_OK)
            return rf;
    }
    if (chain != ctx->chain)
    {
        park_chain(ctx);
        unpark_chain(ctx, chain);
        ctx->pslot[0].p_B = 0; // lvbgpu_proposal_edits names candidates of the selected chain's last batch only
    }
    return LVBGPU_OK;
}

extern "C" int32_t lvbgpu_chains(const lvbgpu_ctx *ctx) { return ctx ? ctx->nchains : 0; }

extern "C" int lvbgpu_set_tree(lvbgpu_ctx *ctx, const int32_t *left, const int32_t *right, int32_t root,
                               int64_t *length_out)
{
    if (!ctx || !left || !right)
        return LVBGPU_E_ARG;
    {
        const int rf = settle(ctx); // the host side of a chain commit still on its way
        if (rf != LVBGPU_OK)
            return rf;
    }
    ENTER(ctx);
    std::string why;
    Topology t;
    if (!t.assign((int32_t)ctx->n, left, right, root, &why))
        return ctx->fail(LVBGPU_E_TOPOLOGY, why);
    ctx->topo = std::move(t);
    ctx->topo_version = ++ctx->version_counter;
    ctx->have_tree = false;
    Program prog;
    ctx->pb.build_full(ctx->topo, prog);
    int rc = run_commit_program(ctx, prog, true, true);
    if (rc != LVBGPU_OK)
        return rc;
    ctx->have_tree = true;
    if (length_out)
        *length_out = ctx->cur_length;
    if (ctx->cur_length <= 0)
        return ctx->fail(LVBGPU_E_ZEROLEN, "assertion failed: changes > 0 (full evaluation gave " + std::to_string(ctx->cur_length) + ")");
    return LVBGPU_OK;
}

extern "C" int lvbgpu_current_length(lvbgpu_ctx *ctx, int64_t *length_out)
{
    if (!ctx || !length_out)
        return LVBGPU_E_ARG;
    if (!ctx->have_tree)
        return ctx->fail(LVBGPU_E_STATE, "no resident tree: call lvbgpu_set_tree first");
    ENTER(ctx);
    const int rc = read_current_length(ctx);
    if (rc != LVBGPU_OK)
        return rc;
    *length_out = ctx->cur_length;
    return LVBGPU_OK;
}

extern "C" int lvbgpu_get_topology(lvbgpu_ctx *ctx, int32_t *parent, int32_t *left, int32_t *right, int32_t *root)
{
    if (!ctx)
        return LVBGPU_E_ARG;
    {
        const int rf = settle(ctx); // the host side of a chain commit still on its way
        if (rf != LVBGPU_OK)
            return rf;
    }
    if (!ctx->have_tree)
        return ctx->fail(LVBGPU_E_STATE, "no resident tree: call lvbgpu_set_tree first");
    if (parent)
        memcpy(parent, ctx->topo.parent.data(), (size_t)ctx->nb * 4);
    if (left)
        memcpy(left, ctx->topo.left.data(), (size_t)ctx->nb * 4);
    if (right)
        memcpy(right, ctx->topo.right.data(), (size_t)ctx->nb * 4);
    if (root)
        *root = ctx->topo.root;
    return LVBGPU_OK;
}

extern "C" int lvbgpu_get_changes(lvbgpu_ctx *ctx, int64_t *changes)
{
    if (!ctx || !changes)
        return LVBGPU_E_ARG;
    if (!ctx->have_tree)
        return ctx->fail(LVBGPU_E_STATE, "no resident tree: call lvbgpu_set_tree first");
    ENTER(ctx);
    memset(changes, 0, (size_t)ctx->n * 8); // leaves hold nothing
    HIPCHK(ctx, hipMemcpyAsync(changes + ctx->n, ctx->d_changes + ctx->row_of((int32_t)ctx->n), (size_t)ctx->chain_rows() * 8,
                               hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return LVBGPU_OK;
}

extern "C" int lvbgpu_get_sets(lvbgpu_ctx *ctx, int32_t node, uint64_t *out)
{
    if (!ctx || !out || node < 0 || node >= ctx->nb)
        return LVBGPU_E_ARG;
    if (node >= ctx->n && !ctx->have_tree)
        return ctx->fail(LVBGPU_E_STATE, "no resident tree: call lvbgpu_set_tree first");
    ENTER(ctx);
    // resident rows are bit planes; hand back the reference's nibble layout
    HIPCHK(ctx, ctx->d_export.reserve((size_t)ctx->stride_words * 8));
    HIPCHK(ctx, launch_export_row((const uint4 *)ctx->d_rows, ctx->row_of(node), ctx->rows_total(), ctx->ntiles,
                                  (uint4 *)ctx->d_export.p, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->d_export.p, (size_t)ctx->nwords * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return LVBGPU_OK;
}

// =================================================================================== batches

extern "C" int lvbgpu_commit(lvbgpu_ctx *ctx, int32_t n_edits, const lvbgpu_edit *edits, int32_t root,
                             int64_t *length_out)
{
    if (!ctx || n_edits < 0 || (n_edits > 0 && !edits))
        return LVBGPU_E_ARG;
    {
        const int rf = settle(ctx); // the host side of a chain commit still on its way
        if (rf != LVBGPU_OK)
            return rf;
    }
    if (!ctx->have_tree)
        return ctx->fail(LVBGPU_E_STATE, "no resident tree: call lvbgpu_set_tree first");
    ENTER(ctx);
    std::string why;
    Program prog;
    if (!ctx->pb.build_candidate(ctx->topo, reinterpret_cast<const Edit *>(edits), n_edits, root, prog, &why))
        return ctx->fail(LVBGPU_E_TOPOLOGY, why);
    if (!ctx->pb.apply_edits(ctx->topo, reinterpret_cast<const Edit *>(edits), n_edits, root, &why))
        return ctx->fail(LVBGPU_E_TOPOLOGY, why);
    ctx->topo_version = ++ctx->version_counter;
    // length_out == NULL: the caller knows the length (it scored this candidate): nothing is
    // read back and nothing waits - the commit is ordered before later work on the stream
    int rc = run_commit_program(ctx, prog, false, length_out != nullptr);
    if (rc != LVBGPU_OK)
    {
        ctx->have_tree = false; // resident state is no longer trustworthy
        return rc;
    }
    if (length_out)
    {
        *length_out = ctx->cur_length;
        if (ctx->cur_length <= 0)
            return ctx->fail(LVBGPU_E_ZEROLEN, "assertion failed: changes > 0 (length after the commit: " + std::to_string(ctx->cur_length) + ")");
    }
    return LVBGPU_OK;
}

extern "C" int lvbgpu_timer_start(lvbgpu_ctx *ctx)
{
    if (!ctx)
        return LVBGPU_E_ARG;
    ENTER(ctx);
    HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return LVBGPU_OK;
}

extern "C" int lvbgpu_timer_stop(lvbgpu_ctx *ctx, float *elapsed_ms)
{
    if (!ctx || !elapsed_ms)
        return LVBGPU_E_ARG;
    ENTER(ctx);
    HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    HIPCHK(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    return LVBGPU_OK;
}

namespace lvbgpu_detail
{
// wait for the timed walks still in flight and add their durations up
int walk_timing_drain(lvbgpu_ctx *ctx)
{
    for (int i = 0; i < ctx->wt_pending; i++)
    {
        float ms = 0.f;
        HIPCHK(ctx, hipEventSynchronize(ctx->wt_ev[2 * i + 1]));
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->wt_ev[2 * i], ctx->wt_ev[2 * i + 1]));
        ctx->wt_ms += ms;
        ctx->wt_launches++;
    }
    ctx->wt_pending = 0;
    return LVBGPU_OK;
}
} // namespace lvbgpu_detail

extern "C" int lvbgpu_walk_timing(lvbgpu_ctx *ctx, int enable)
{
    if (!ctx || enable < 0)
        return LVBGPU_E_ARG;
    ENTER(ctx);
    if (enable)
    {
        for (hipEvent_t &ev : ctx->wt_ev)
            if (!ev)
                HIPCHK(ctx, hipEventCreate(&ev));
        ctx->wt_pending = 0;
        ctx->wt_ms = 0.0;
        ctx->wt_launches = 0;
        ctx->wt_every = (uint32_t)enable;
        ctx->wt_seen = 0;
        ctx->walk_timing = true;
        return LVBGPU_OK;
    }
    const int rc = ctx->walk_timing ? walk_timing_drain(ctx) : LVBGPU_OK;
    ctx->walk_timing = false;
    return rc;
}

extern "C" int lvbgpu_walk_timing_read(lvbgpu_ctx *ctx, double *total_ms, int64_t *launches)
{
    if (!ctx || !total_ms || !launches)
        return LVBGPU_E_ARG;
    ENTER(ctx);
    const int rc = walk_timing_drain(ctx);
    if (rc != LVBGPU_OK)
        return rc;
    *total_ms = ctx->wt_ms;
    *launches = ctx->wt_launches;
    return LVBGPU_OK;
}

extern "C" int lvbgpu_probe_l2(lvbgpu_ctx *ctx, int32_t B, int32_t rows_per_wave, int32_t reps, double *gb_per_s)
{
    if (!ctx || !gb_per_s || B < 1 || rows_per_wave < 8 || reps < 1 || (uint64_t)B * ctx->ntiles >= (1ull << 31))
        return LVBGPU_E_ARG;
    ENTER(ctx);
    HIPCHK(ctx, ctx->d_probe_sink.reserve(64));
    const uint32_t ngroups = choose_groups((uint32_t)B, ctx->ntiles, ctx->target_waves);
    double best = 0.0;
    // the resident block is tile-major (stride 64 in the probe kernel); LVBGPU_PROBE_ROW_MAJOR: the same reads as if it
    // were row-major, as it was until round 3 (tools/cfg5_probe.py)
    const uint32_t probe_stride4 = getenv("LVBGPU_PROBE_ROW_MAJOR") ? ctx->stride4 : 64u;
    for (int ring : {4, 8})
    {
        uint64_t loads = 0;
        for (int i = 0; i < 3; i++) // warm the caches and the clocks
            HIPCHK(ctx, launch_l2_probe((const uint4 *)ctx->d_rows, probe_stride4, ctx->rows_total(), ctx->ntiles, ngroups,
                                        (uint32_t)B, (uint32_t)rows_per_wave, ring, (uint4 *)ctx->d_probe_sink.p, &loads,
                                        ctx->stream));
        HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        for (int i = 0; i < reps; i++)
            HIPCHK(ctx, launch_l2_probe((const uint4 *)ctx->d_rows, probe_stride4, ctx->rows_total(), ctx->ntiles, ngroups,
                                        (uint32_t)B, (uint32_t)rows_per_wave, ring, (uint4 *)ctx->d_probe_sink.p, &loads,
                                        ctx->stream));
        HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
        float ms = 0.f;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        if (ms > 0.f)
            best = std::max(best, (double)loads * 1024.0 * reps / (ms * 1e-3) / 1e9);
        if (getenv("LVBGPU_PROBE_VERBOSE") && ms > 0.f) // both depths, not only the better one (tools/cfg5_probe.py)
            fprintf(stderr, "[lvbgpu_probe_l2] B=%d rows/wave=%d ring=%d: %.1f us per launch, %.0f GB/s\n", B, rows_per_wave, ring,
                    1e3 * ms / reps, (double)loads * 1024.0 * reps / (ms * 1e-3) / 1e9);
    }
    *gb_per_s = best;
    return LVBGPU_OK;
}

extern "C" int lvbgpu_synchronize(lvbgpu_ctx *ctx)
{
    if (!ctx)
        return LVBGPU_E_ARG;
    {
        const int rf = settle(ctx); // the host side of a chain commit still on its way
        if (rf != LVBGPU_OK)
            return rf;
    }
    ENTER(ctx);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));
    return LVBGPU_OK;
}

extern "C" void *lvbgpu_stream(lvbgpu_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" int lvbgpu_set_wait_limit(lvbgpu_ctx *ctx, double seconds)
{
    if (!ctx || !(seconds > 0.0))
        return LVBGPU_E_ARG;
    ctx->wait_limit_s = seconds;
    return LVBGPU_OK;
}

// test hook for the wait limit: keeps the context's stream busy for about `ms` milliseconds (bounded: <= 2000) with a
// kernel that does nothing but watch the clock, so that a step enqueued behind it cannot complete before then
// test hook: counters of things a test cannot see from results (results are the same either way)
extern "C" int lvbgpu_debug_count(lvbgpu_ctx *ctx, int32_t what, int64_t *count)
{
    if (!ctx || !count)
        return LVBGPU_E_ARG;
    switch (what)
    {
    case LVBGPU_COUNT_PAIRED_WALKS: *count = ctx->paired_walks; break;
    case LVBGPU_COUNT_COMMITS_REUSING_PROGRAMS: *count = ctx->commits_reusing_programs; break;
    case LVBGPU_COUNT_POST_LAUNCHES: *count = ctx->post_launches; break;
    case LVBGPU_COUNT_POST_LAUNCHES_WITH_GENERATOR: *count = ctx->post_launches_with_generator; break;
    default: return LVBGPU_E_ARG;
    }
    return LVBGPU_OK;
}

extern "C" int lvbgpu_debug_stall(lvbgpu_ctx *ctx, int32_t ms)
{
    if (!ctx || ms < 1 || ms > 2000)
        return LVBGPU_E_ARG;
    ENTER(ctx);
    HIPCHK(ctx, launch_stall(ctx->stream, (uint32_t)ms));
    return LVBGPU_OK;
}

