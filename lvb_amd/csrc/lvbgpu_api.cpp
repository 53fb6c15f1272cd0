// lvbgpu_api.cpp - the C ABI of include/lvbgpu.h over the HIP kernels (fitch_kernels.hip) and
// the host-side program builder (program.cpp).  Compiled by hipcc into lvb_amd/liblvbgpu.so.
//
// There is deliberately no CPU implementation of any scoring entry point in this file: without
// a working HIP device every one of them fails with LVBGPU_E_NODEVICE / LVBGPU_E_HIP.
#include "../../include/lvbgpu.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "kernels.hpp"
#include "pool.hpp"
#include "program.hpp"

using namespace lvbgpu;

namespace
{

constexpr int ABI_VERSION = 1;

inline uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

// growable device buffer
struct DevBuf
{
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap)
            return hipSuccess;
        if (p)
            (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = std::max(bytes, (size_t)4096);
        want = want + want / 4;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess)
            cap = want;
        return e;
    }
    void release()
    {
        if (p)
            (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// growable pinned host buffer
struct PinBuf
{
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap)
            return hipSuccess;
        if (p)
            (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = std::max(bytes, (size_t)4096);
        want = want + want / 4;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess)
            cap = want;
        return e;
    }
    void release()
    {
        if (p)
            (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

// the reference's node record (LVB.h:121-128), as the strict-compat entry sees it
struct RefNode
{
    long parent, left, right, changes;
    uint64_t *sitestate;
};
static_assert(sizeof(RefNode) == 40, "reference node record is 40 bytes");

// RCCL is loaded on demand so that the scoring library does not depend on it at load time
struct Id128
{
    char bytes[128]; // ncclUniqueId
};
struct BuildWorker
{
    Topology topo;
    uint64_t topo_version = ~0ull;
    ProgramBuilder pb;
    Program prog;
    std::vector<CandDesc> cands;
    int32_t max_stack = 0;
    int64_t dirty = 0;
    int rc = 0;
    std::string why;
};

struct Rccl
{
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, Id128 /* ncclUniqueId by value */, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
} // namespace

struct lvbgpu_ctx
{
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    long n = 0, nwords = 0;
    int32_t nb = 0;
    uint32_t stride_words = 0, stride4 = 0, ntiles = 0;
    uint32_t target_waves = TARGET_WAVES; // tuning knob (env LVBGPU_TARGET_WAVES)

    uint64_t *d_rows = nullptr;              // [nb][stride_words]
    unsigned long long *d_changes = nullptr; // [nb + 1]; slot nb = the two root combines
    long long *d_scalars = nullptr;          // [0] S_all, [1] current length
    bool have_tree = false;
    int64_t cur_length = 0;

    Topology topo;
    uint64_t topo_version = 0; // bumped whenever topo changes (workers keep private copies)
    ProgramBuilder pb;
    Pool *pool = nullptr;
    std::vector<BuildWorker> workers;

    DevBuf d_len; // length slot of single-program launches (set_tree, commit)
    static constexpr int COMMIT_SLOTS = 4;
    PinBuf h_commit[COMMIT_SLOTS]; // commit programs in flight (asynchronous commits)
    DevBuf d_commit[COMMIT_SLOTS];
    hipEvent_t commit_ev[COMMIT_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    int commit_slot = 0;
    bool cur_length_stale = false; // the device holds a newer length than cur_length
    DevBuf d_export;      // one row in nibble layout (lvbgpu_get_sets)
    lvbgpu_batch *step_batch = nullptr; // recycled by lvbgpu_score_batch
    lvbgpu_batch *full_batch = nullptr; // recycled by lvbgpu_score_full_batch
    // device-side proposals (lvbgpu_propose_score)
    lvbgpu_batch *prop_batch = nullptr;
    DevBuf d_topo4, d_pedits, d_pinfo;
    PinBuf h_pinfo;
    DevBuf d_moves; // moves named by the host (lvbgpu_score_moves)
    PinBuf h_moves;
    uint64_t d_topo_version = ~0ull;
    uint32_t p_stride_t = 0, p_stride_e = 0;
    int32_t p_B = 0; // candidates of the last device batch (0: none)
    PinBuf h_pin;
    DevBuf d_cin, d_cout; // strict-compat arenas
    PinBuf h_cin, h_cout;
    std::vector<int32_t> slot_of;
    std::vector<uint32_t> slot_epoch;
    uint32_t slot_gen = 0;

    void *comm = nullptr;
    int comm_rank = 0, comm_size = 1;
    DevBuf d_comm;

    std::string last_error;

    int fail_hip(hipError_t e, const char *what)
    {
        last_error = std::string(what) + ": " + hipGetErrorString(e);
        return (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver)
                   ? LVBGPU_E_NODEVICE
                   : (e == hipErrorOutOfMemory ? LVBGPU_E_NOMEM : LVBGPU_E_HIP);
    }
    int fail(int code, const std::string &why)
    {
        last_error = why;
        return code;
    }
};

struct lvbgpu_batch
{
    lvbgpu_ctx *ctx = nullptr;
    int32_t B = 0;
    DevBuf d_prog; // [cands][toks][dsts]
    DevBuf d_len;
    PinBuf h_len;  // lengths land here after every launch (async copy on the context's stream)
    size_t off_toks = 0, off_dsts = 0;
    lvbgpu_batch_stats stats{};
    bool full_mode = false; // whole topologies: reads leaf rows only
    // step batches owned by the context (lvbgpu_score_batch / _score_full_batch) run build -> launch
    // -> lengths back to back, which lets them drop one synchronisation and take the zeroing of the
    // length slots off the critical path
    bool recycled = false;
    bool len_zeroed = false; // d_len was cleared after the previous read-back
};

// steps up to this many candidates finish within a few hundred microseconds: poll for them instead of
// sleeping in the runtime (its wake-up costs ~10 us per step)
constexpr int32_t SPIN_WAIT_MAX_B = 16384;

#define HIPCHK(ctx, call)                                                                                              \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e__ = (call);                                                                                       \
        if (e__ != hipSuccess)                                                                                         \
            return (ctx)->fail_hip(e__, #call);                                                                        \
    } while (0)

namespace
{

static Rccl g_rccl;
static thread_local std::string g_last_error_noctx;

int hip_status_noctx(hipError_t e, const char *what)
{
    g_last_error_noctx = std::string(what) + ": " + hipGetErrorString(e);
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver)
               ? LVBGPU_E_NODEVICE
               : (e == hipErrorOutOfMemory ? LVBGPU_E_NOMEM : LVBGPU_E_HIP);
}

// pack programs -> one host blob [CandDesc x B][toks][dsts] (16-byte aligned sections)
struct Packed
{
    std::vector<CandDesc> cands;
    std::vector<uint32_t> toks;
    std::vector<int32_t> dsts;
    int32_t max_stack = 0;
    int64_t dirty = 0;
    void add(const Program &p, size_t tok0, size_t dst0, long long base, uint32_t flags)
    {
        CandDesc cd{};
        cd.tok_off = (uint32_t)tok0;
        cd.ntok = (uint32_t)(p.toks.size() - tok0);
        cd.dst_off = (uint32_t)dst0;
        cd.ncomb = (uint32_t)(p.dsts.size() - dst0);
        cd.base = base;
        cd.flags = flags;
        for (size_t i = tok0; i < p.toks.size(); i++)
            cd.nfresh += (p.toks[i] & TOK_FRESH) ? 1u : 0u;
        cands.push_back(cd);
    }
};

size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

// end of a search step on the context's stream
hipError_t wait_for_step(lvbgpu_ctx *ctx, int32_t B)
{
    if (B > SPIN_WAIT_MAX_B)
        return hipStreamSynchronize(ctx->stream);
    hipError_t q;
    while ((q = hipStreamQuery(ctx->stream)) == hipErrorNotReady)
        ;
    return q;
}

WalkArgs resident_args(lvbgpu_ctx *ctx, const DevBuf &prog, size_t off_toks, size_t off_dsts, void *d_len,
                       uint32_t B, int32_t max_stack)
{
    WalkArgs a{};
    a.rows_in = (const uint4 *)ctx->d_rows;
    a.rows_out = (uint4 *)ctx->d_rows;
    a.cands = (const CandDesc *)prog.p;
    a.toks = (const uint32_t *)((const char *)prog.p + off_toks);
    a.dsts = (const int32_t *)((const char *)prog.p + off_dsts);
    a.node_changes = (const long long *)ctx->d_changes;
    a.s_all = ctx->d_scalars;
    a.len_out = (unsigned long long *)d_len;
    a.changes_out = ctx->d_changes;
    a.in_stride4 = ctx->stride4;
    a.out_stride4 = ctx->stride4;
    a.B = B;
    a.ntiles = ctx->ntiles;
    a.ngroups = choose_groups(B, ctx->ntiles, ctx->target_waves);
    a.nitems = B * a.ngroups;
    a.stack_depth = (uint32_t)std::max(max_stack, 1);
    a.root_slot = (uint32_t)ctx->nb;
    return a;
}

int check_depth(lvbgpu_ctx *ctx, int32_t max_stack)
{
    if ((size_t)max_stack * WALK_WAVES * 64 * sizeof(uint4) > MAX_LDS_BYTES)
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
    // tuning knobs for experiments: extra row padding (in tiles) and the wave-count target
    const char *pad = getenv("LVBGPU_STRIDE_PAD_TILES");
    ctx->stride_words = (ctx->ntiles + (pad ? (uint32_t)atoi(pad) : 0u)) * TILE_WORDS;
    ctx->stride4 = ctx->stride_words / 2;
    if (const char *tw = getenv("LVBGPU_TARGET_WAVES"))
        ctx->target_waves = (uint32_t)std::max(1, atoi(tw));
    // (row offsets are 64-bit in the kernels: the tree block is limited by HBM, not by index width)
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIPCHK(ctx, hipEventCreate(&ctx->ev0));
    HIPCHK(ctx, hipEventCreate(&ctx->ev1));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_rows, (size_t)ctx->nb * ctx->stride_words * 8));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_changes, (size_t)(ctx->nb + 1) * 8));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_scalars, 2 * 8));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_changes, 0, (size_t)(ctx->nb + 1) * 8, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_scalars, 0, 16, ctx->stream));
    HIPCHK(ctx, upload_iupac_table());
    HIPCHK(ctx, raise_lds_limit());
    ctx->pb.resize(ctx->nb);
    return LVBGPU_OK;
}

// everything that is not a leaf word becomes all-ones (inert under fitch); then the leaf rows go
// from the reference's nibble layout to the device's bit-plane layout (all-ones stays all-ones)
int finish_rows(lvbgpu_ctx *ctx)
{
    HIPCHK(ctx, launch_fill_pad(ctx->d_rows, (uint32_t)ctx->nb, (uint32_t)ctx->nwords, ctx->stride_words,
                                (uint32_t)ctx->n, ctx->stream));
    HIPCHK(ctx, launch_relayout((uint4 *)ctx->d_rows, (uint32_t)ctx->n, ctx->stride4, true, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return LVBGPU_OK;
}

} // namespace

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

namespace
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
} // namespace

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
    if (ctx->d_rows)
        (void)hipFree(ctx->d_rows);
    if (ctx->d_changes)
        (void)hipFree(ctx->d_changes);
    if (ctx->d_scalars)
        (void)hipFree(ctx->d_scalars);
    delete ctx->pool;
    ctx->pool = nullptr;
    ctx->d_topo4.release();
    ctx->d_pedits.release();
    ctx->d_pinfo.release();
    ctx->h_pinfo.release();
    ctx->d_moves.release();
    ctx->h_moves.release();
    for (lvbgpu_batch *rb : {ctx->step_batch, ctx->full_batch, ctx->prop_batch})
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
    if (ctx->ev0)
        (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1)
        (void)hipEventDestroy(ctx->ev1);
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
    if (rc == LVBGPU_OK)
    {
        hipError_t e = hipMemcpy2DAsync(ctx->d_rows, (size_t)ctx->stride_words * 8, leaf_matrix,
                                        (size_t)row_stride_words * 8, (size_t)nwords * 8, (size_t)n,
                                        hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess)
            rc = ctx->fail_hip(e, "upload leaf matrix");
    }
    if (rc == LVBGPU_OK)
        rc = finish_rows(ctx);
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
    if (rc == LVBGPU_OK)
        rc = encode_into(ctx, n, m, rows, ctx->d_rows, ctx->stride_words);
    if (rc == LVBGPU_OK)
        rc = finish_rows(ctx);
    if (rc != LVBGPU_OK)
    {
        g_last_error_noctx = ctx->last_error;
        lvbgpu_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return LVBGPU_OK;
}

extern "C" long lvbgpu_n(const lvbgpu_ctx *ctx) { return ctx ? ctx->n : 0; }
extern "C" long lvbgpu_nwords(const lvbgpu_ctx *ctx) { return ctx ? ctx->nwords : 0; }

// =================================================================================== resident tree

namespace
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
    HIPCHK(ctx, hipMemcpyAsync(&s_all, ctx->d_scalars, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&root_changes, ctx->d_changes + ctx->nb, 8, hipMemcpyDeviceToHost, ctx->stream));
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
    pk.add(prog, 0, 0, 0, 0);
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
    HIPCHK(ctx, hipMemcpyAsync(dprog.p, h, total, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->commit_ev[slot], ctx->stream));
    HIPCHK(ctx, ctx->d_len.reserve(8));
    if (zero_all)
    {
        HIPCHK(ctx, hipMemsetAsync(ctx->d_len.p, 0, 8, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(ctx->d_changes, 0, (size_t)(ctx->nb + 1) * 8, ctx->stream));
    }
    else // the accept path: one small launch clears everything the walk accumulates into and takes the
         // recomputed nodes' old counts out of S_all
        HIPCHK(ctx, launch_zero_changes((unsigned long long *)ctx->d_changes, (const int32_t *)((const char *)dprog.p + o_d),
                                        (uint32_t)prog.dsts.size(), (unsigned long long *)ctx->d_changes + ctx->nb,
                                        (unsigned long long *)ctx->d_len.p, (unsigned long long *)ctx->d_scalars,
                                        ctx->stream));
    WalkArgs a = resident_args(ctx, dprog, o_t, o_d, ctx->d_len.p, 1, prog.max_stack);
    if (!zero_all)
        a.s_all_out = (unsigned long long *)ctx->d_scalars; // the walk adds the new counts: S_all stays current
    HIPCHK(ctx, launch_walk(a, true, ctx->stream));
    if (zero_all) // full evaluation: sum once
        HIPCHK(ctx, launch_sum_changes(ctx->d_changes, (uint32_t)ctx->n, (uint32_t)ctx->nb, ctx->d_scalars, ctx->stream));
    ctx->cur_length_stale = true;
    return readback ? read_current_length(ctx) : LVBGPU_OK;
}
} // namespace

extern "C" int lvbgpu_set_tree(lvbgpu_ctx *ctx, const int32_t *left, const int32_t *right, int32_t root,
                               int64_t *length_out)
{
    if (!ctx || !left || !right)
        return LVBGPU_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::string why;
    Topology t;
    if (!t.assign((int32_t)ctx->n, left, right, root, &why))
        return ctx->fail(LVBGPU_E_TOPOLOGY, why);
    ctx->topo = std::move(t);
    ctx->topo_version++;
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
        return ctx->fail(LVBGPU_E_ZEROLEN, "assertion failed: changes > 0");
    return LVBGPU_OK;
}

extern "C" int lvbgpu_current_length(lvbgpu_ctx *ctx, int64_t *length_out)
{
    if (!ctx || !length_out)
        return LVBGPU_E_ARG;
    if (!ctx->have_tree)
        return ctx->fail(LVBGPU_E_STATE, "no resident tree: call lvbgpu_set_tree first");
    HIPCHK(ctx, hipSetDevice(ctx->device));
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
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(changes, ctx->d_changes, (size_t)ctx->nb * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return LVBGPU_OK;
}

extern "C" int lvbgpu_get_sets(lvbgpu_ctx *ctx, int32_t node, uint64_t *out)
{
    if (!ctx || !out || node < 0 || node >= ctx->nb)
        return LVBGPU_E_ARG;
    if (node >= ctx->n && !ctx->have_tree)
        return ctx->fail(LVBGPU_E_STATE, "no resident tree: call lvbgpu_set_tree first");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // resident rows are bit planes; hand back the reference's nibble layout
    HIPCHK(ctx, ctx->d_export.reserve((size_t)ctx->stride_words * 8));
    HIPCHK(ctx, launch_export_row((const uint4 *)(ctx->d_rows + (size_t)node * ctx->stride_words),
                                  (uint4 *)ctx->d_export.p, ctx->stride4, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->d_export.p, (size_t)ctx->nwords * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return LVBGPU_OK;
}

// =================================================================================== batches

extern "C" void lvbgpu_batch_free(lvbgpu_batch *b)
{
    if (!b)
        return;
    if (b->ctx)
    {
        (void)hipSetDevice(b->ctx->device);
        (void)hipStreamSynchronize(b->ctx->stream);
    }
    b->d_prog.release();
    b->d_len.release();
    b->h_len.release();
    delete b;
}


namespace
{
constexpr int32_t PARALLEL_BUILD_MIN = 512; // below this one thread is faster than waking the pool

// what a batch is made from: edits against the resident tree, or whole topologies
struct BuildJob
{
    const int32_t *edit_offsets = nullptr;
    const lvbgpu_edit *edits = nullptr;
    const int32_t *roots = nullptr;
    const int32_t *left = nullptr, *right = nullptr; // full mode: [B][2n-3]
    bool full = false;
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
    if (w.topo_version != ctx->topo_version)
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
        cd.flags = CAND_RESIDENT_BASE;
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
    const int32_t par_min = job.full ? 16 : PARALLEL_BUILD_MIN;
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
    HIPCHK(ctx, ctx->h_pin.reserve(total));
    char *h = (char *)ctx->h_pin.p;
    auto gather = [&](int t) {
        BuildWorker &w = ctx->workers[t];
        CandDesc *cd = (CandDesc *)h + cand_base[t];
        for (size_t i = 0; i < w.cands.size(); i++)
        {
            cd[i] = w.cands[i];
            cd[i].tok_off += (uint32_t)tok_base[t];
            cd[i].dst_off += (uint32_t)dst_base[t];
        }
        memcpy(h + o_t + tok_base[t] * 4, w.prog.toks.data(), w.prog.toks.size() * 4);
        memcpy(h + o_d + dst_base[t] * 4, w.prog.dsts.data(), w.prog.dsts.size() * 4);
    };
    if (T == 1)
        gather(0);
    else
        ctx->pool->run(T, gather);
    HIPCHK(ctx, hipMemcpyAsync(bt->d_prog.p, h, total, hipMemcpyHostToDevice, ctx->stream));
    // h_pin is reused by the next upload.  A recycled step batch is read back (and the stream
    // drained) by lvbgpu_batch_lengths before anything can build again: no need to wait here.
    if (!bt->recycled)
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    bt->ctx = ctx;
    bt->B = B;
    bt->off_toks = o_t;
    bt->off_dsts = o_d;
    const void *old_len = bt->d_len.p;
    HIPCHK(ctx, bt->d_len.reserve((size_t)B * 8));
    if (bt->d_len.p != old_len)
        bt->len_zeroed = false;
    HIPCHK(ctx, bt->h_len.reserve((size_t)B * 8));
    bt->full_mode = job.full;
    bt->stats.candidates = B;
    bt->stats.combines = (int64_t)ndst;
    bt->stats.rows_read = (int64_t)ntok;
    bt->stats.dirty_nodes = dirty;
    bt->stats.max_stack = max_stack;
    bt->stats.algorithmic_bytes = bt->stats.rows_read * ctx->nwords * 8;
    return LVBGPU_OK;
}
} // namespace

extern "C" int lvbgpu_batch_build(lvbgpu_ctx *ctx, int32_t B, const int32_t *edit_offsets, const lvbgpu_edit *edits,
                                  const int32_t *roots, lvbgpu_batch **out)
{
    if (!ctx || !out || B < 1 || !edit_offsets || (!edits && edit_offsets[B] > 0))
        return LVBGPU_E_ARG;
    *out = nullptr;
    if (!ctx->have_tree)
        return ctx->fail(LVBGPU_E_STATE, "no resident tree: call lvbgpu_set_tree first");
    HIPCHK(ctx, hipSetDevice(ctx->device));
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
    *out = bt;
    return LVBGPU_OK;
}

extern "C" int lvbgpu_batch_launch(lvbgpu_ctx *ctx, lvbgpu_batch *b)
{
    if (!ctx || !b || b->ctx != ctx)
        return LVBGPU_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!b->len_zeroed)
        HIPCHK(ctx, hipMemsetAsync(b->d_len.p, 0, (size_t)b->B * 8, ctx->stream));
    b->len_zeroed = false;
    WalkArgs a = resident_args(ctx, b->d_prog, b->off_toks, b->off_dsts, b->d_len.p, (uint32_t)b->B,
                               (int32_t)b->stats.max_stack);
    HIPCHK(ctx, launch_walk(a, false, ctx->stream));
    return LVBGPU_OK;
}

extern "C" int lvbgpu_batch_lengths(lvbgpu_ctx *ctx, lvbgpu_batch *b, int64_t *lengths_out)
{
    if (!ctx || !b || b->ctx != ctx || !lengths_out)
        return LVBGPU_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // through pinned memory: one DMA, no staging
    HIPCHK(ctx, hipMemcpyAsync(b->h_len.p, b->d_len.p, (size_t)b->B * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (b->recycled)
        HIPCHK(ctx, wait_for_step(ctx, b->B));
    else
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (b->recycled)
    {
        // clear the slots for the next step now, while the host consumes these lengths
        HIPCHK(ctx, hipMemsetAsync(b->d_len.p, 0, b->d_len.cap, ctx->stream));
        b->len_zeroed = true;
    }
    memcpy(lengths_out, b->h_len.p, (size_t)b->B * 8);
    for (int32_t i = 0; i < b->B; i++)
        if (lengths_out[i] <= 0)
            return ctx->fail(LVBGPU_E_ZEROLEN, "assertion failed: changes > 0");
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
    if (!ctx->have_tree)
        return ctx->fail(LVBGPU_E_STATE, "no resident tree: call lvbgpu_set_tree first");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // one recycled batch per context: a search calls this every step, so no allocation here
    if (!ctx->step_batch)
    {
        ctx->step_batch = new (std::nothrow) lvbgpu_batch();
        if (!ctx->step_batch)
            return LVBGPU_E_NOMEM;
        ctx->step_batch->recycled = true;
    }
    lvbgpu_batch *b = ctx->step_batch;
    BuildJob job;
    job.edit_offsets = edit_offsets;
    job.edits = edits;
    job.roots = roots;
    int rc = build_into(ctx, b, B, job);
    if (rc == LVBGPU_OK)
        rc = lvbgpu_batch_launch(ctx, b);
    if (rc == LVBGPU_OK)
        rc = lvbgpu_batch_lengths(ctx, b, lengths_out);
    return rc;
}

// ---- device-side neighbourhoods

namespace
{
// parent | left | right | number of leaves below, for propose_kernel
int sync_device_topology(lvbgpu_ctx *ctx)
{
    if (ctx->d_topo_version == ctx->topo_version)
        return LVBGPU_OK;
    const int32_t nb = ctx->nb;
    const Topology &t = ctx->topo;
    std::vector<int32_t> host((size_t)4 * nb);
    memcpy(host.data(), t.parent.data(), (size_t)nb * 4);
    memcpy(host.data() + nb, t.left.data(), (size_t)nb * 4);
    memcpy(host.data() + 2 * (size_t)nb, t.right.data(), (size_t)nb * 4);
    int32_t *leaves = host.data() + 3 * (size_t)nb;
    // leaves below each node: children before parents via an explicit preorder
    std::vector<int32_t> order;
    order.reserve(nb);
    std::vector<int32_t> st{t.root};
    while (!st.empty())
    {
        const int32_t v = st.back();
        st.pop_back();
        order.push_back(v);
        if (t.left[v] >= 0)
        {
            st.push_back(t.left[v]);
            st.push_back(t.right[v]);
        }
    }
    for (auto it = order.rbegin(); it != order.rend(); ++it)
    {
        const int32_t v = *it;
        leaves[v] = (t.left[v] < 0 || v == t.root) ? 1 : leaves[t.left[v]] + leaves[t.right[v]];
    }
    HIPCHK(ctx, ctx->d_topo4.reserve(host.size() * 4));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_topo4.p, host.data(), host.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); // `host` goes out of scope
    ctx->d_topo_version = ctx->topo_version;
    return LVBGPU_OK;
}
} // namespace

namespace
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
    const void *old_len = bt->d_len.p;
    HIPCHK(ctx, bt->d_len.reserve((size_t)B * 8));
    if (bt->d_len.p != old_len)
        bt->len_zeroed = false;
    HIPCHK(ctx, bt->h_len.reserve((size_t)B * 8));
    HIPCHK(ctx, ctx->d_pedits.reserve((size_t)B * stride_e * sizeof(lvbgpu_edit_dev)));
    HIPCHK(ctx, ctx->d_pinfo.reserve((size_t)B * sizeof(ProposalInfo)));
    bt->ctx = ctx;
    bt->B = B;
    bt->off_toks = o_t;
    bt->off_dsts = o_d;
    bt->full_mode = false;
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
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_moves.p, ctx->h_moves.p, (size_t)B * sizeof(lvbgpu_move), hipMemcpyHostToDevice,
                                   ctx->stream));
        d_moves = (const lvbgpu_move_dev *)ctx->d_moves.p;
    }
    HIPCHK(ctx, launch_propose((const int32_t *)ctx->d_topo4.p, (int32_t)ctx->n, ctx->topo.root, kind, mix_a, mix_b, seed,
                               (uint32_t)B,
                               stride_t, stride_e, (uint32_t *)((char *)bt->d_prog.p + o_t),
                               (int32_t *)((char *)bt->d_prog.p + o_d), (lvbgpu_edit_dev *)ctx->d_pedits.p,
                               (CandDesc *)bt->d_prog.p, (ProposalInfo *)ctx->d_pinfo.p, d_moves, ctx->stream));
    rc = lvbgpu_batch_launch(ctx, bt);
    if (rc != LVBGPU_OK)
        return rc;
    // only the lengths come back per step; a move's descriptor and edits are fetched when (and only
    // when) the caller wants that candidate (lvbgpu_proposal_edits)
    HIPCHK(ctx, hipMemcpyAsync(bt->h_len.p, bt->d_len.p, (size_t)B * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, wait_for_step(ctx, B));
    HIPCHK(ctx, hipMemsetAsync(bt->d_len.p, 0, bt->d_len.cap, ctx->stream)); // for the next step, off its critical path
    bt->len_zeroed = true;
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
            return ctx->fail(LVBGPU_E_ZEROLEN, "assertion failed: changes > 0");
    }
    ctx->p_B = B;
    return LVBGPU_OK;
}
} // namespace

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

extern "C" int lvbgpu_score_full_batch(lvbgpu_ctx *ctx, int32_t B, const int32_t *left, const int32_t *right,
                                       const int32_t *roots, int64_t *lengths_out)
{
    if (!ctx || B < 1 || !left || !right || !lengths_out)
        return LVBGPU_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
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

extern "C" int lvbgpu_commit(lvbgpu_ctx *ctx, int32_t n_edits, const lvbgpu_edit *edits, int32_t root,
                             int64_t *length_out)
{
    if (!ctx || n_edits < 0 || (n_edits > 0 && !edits))
        return LVBGPU_E_ARG;
    if (!ctx->have_tree)
        return ctx->fail(LVBGPU_E_STATE, "no resident tree: call lvbgpu_set_tree first");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::string why;
    Program prog;
    if (!ctx->pb.build_candidate(ctx->topo, reinterpret_cast<const Edit *>(edits), n_edits, root, prog, &why))
        return ctx->fail(LVBGPU_E_TOPOLOGY, why);
    if (!ctx->pb.apply_edits(ctx->topo, reinterpret_cast<const Edit *>(edits), n_edits, root, &why))
        return ctx->fail(LVBGPU_E_TOPOLOGY, why);
    ctx->topo_version++;
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
            return ctx->fail(LVBGPU_E_ZEROLEN, "assertion failed: changes > 0");
    }
    return LVBGPU_OK;
}

// =================================================================================== strict compat

extern "C" int lvbgpu_getplen_compat(lvbgpu_ctx *ctx, void *tree_v, long root, int64_t *length_out)
{
    if (!ctx || !tree_v || !length_out)
        return LVBGPU_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
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
    a.in_stride4 = Wp / 2;
    a.out_stride4 = Wp / 2;
    a.B = 1;
    a.ntiles = ctx->ntiles;
    a.ngroups = ctx->ntiles;
    a.nitems = ctx->ntiles;
    a.stack_depth = (uint32_t)std::max(prog.max_stack, 1);
    HIPCHK(ctx, launch_walk(a, true, ctx->stream));
    HIPCHK(ctx, launch_relayout((uint4 *)((char *)ctx->d_cout.p + oo_rows), n_out, Wp / 2, false, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_cout.p, ctx->d_cout.p, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));

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

extern "C" int lvbgpu_timer_start(lvbgpu_ctx *ctx)
{
    if (!ctx)
        return LVBGPU_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return LVBGPU_OK;
}

extern "C" int lvbgpu_timer_stop(lvbgpu_ctx *ctx, float *elapsed_ms)
{
    if (!ctx || !elapsed_ms)
        return LVBGPU_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    HIPCHK(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    return LVBGPU_OK;
}

extern "C" int lvbgpu_synchronize(lvbgpu_ctx *ctx)
{
    if (!ctx)
        return LVBGPU_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return LVBGPU_OK;
}

extern "C" void *lvbgpu_stream(lvbgpu_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

// =================================================================================== RCCL

namespace
{
bool load_rccl(std::string *why)
{
    if (g_rccl.lib)
        return true;
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *nm : names)
        if ((g_rccl.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL)))
            break;
    if (!g_rccl.lib)
    {
        *why = std::string("dlopen librccl.so: ") + dlerror();
        return false;
    }
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(g_rccl.lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(g_rccl.lib, "ncclCommInitRank");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(g_rccl.lib, "ncclAllReduce");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(g_rccl.lib, "ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(g_rccl.lib, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy)
    {
        *why = "librccl.so lacks the nccl* entry points";
        return false;
    }
    return true;
}
constexpr int NCCL_INT64 = 4; // ncclInt64
constexpr int NCCL_MIN = 3;   // ncclMin
} // namespace

extern "C" int lvbgpu_comm_unique_id(void *id128)
{
    if (!id128)
        return LVBGPU_E_ARG;
    std::string why;
    if (!load_rccl(&why))
    {
        g_last_error_noctx = why;
        return LVBGPU_E_COMM;
    }
    const int r = g_rccl.GetUniqueId(id128);
    if (r != 0)
    {
        g_last_error_noctx = std::string("ncclGetUniqueId: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
        return LVBGPU_E_COMM;
    }
    return LVBGPU_OK;
}

extern "C" int lvbgpu_comm_init(lvbgpu_ctx *ctx, int nranks, int rank, const void *id128)
{
    if (!ctx || !id128 || nranks < 1 || rank < 0 || rank >= nranks)
        return LVBGPU_E_ARG;
    std::string why;
    if (!load_rccl(&why))
        return ctx->fail(LVBGPU_E_COMM, why);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Id128 id;
    memcpy(id.bytes, id128, 128);
    const int r = g_rccl.CommInitRank(&ctx->comm, nranks, id, rank);
    if (r != 0)
        return ctx->fail(LVBGPU_E_COMM,
                         std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"));
    ctx->comm_rank = rank;
    ctx->comm_size = nranks;
    HIPCHK(ctx, ctx->d_comm.reserve(16));
    return LVBGPU_OK;
}

extern "C" int lvbgpu_allreduce_min(lvbgpu_ctx *ctx, int64_t *value, int32_t *argmin_rank)
{
    if (!ctx || !value)
        return LVBGPU_E_ARG;
    if (!ctx->comm)
        return ctx->fail(LVBGPU_E_STATE, "no communicator: call lvbgpu_comm_init first");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // one 8-byte min over xGMI finds the best length; a second one over (length, rank) keys
    // names a rank that holds it.  Lengths are < 2^47 (MAX_M * 2 * MAX_N), ranks < 2^16.
    long long vals[2] = {(long long)*value, ((long long)*value << 16) | (long long)ctx->comm_rank};
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_comm.p, vals, 16, hipMemcpyHostToDevice, ctx->stream));
    const int r = g_rccl.AllReduce(ctx->d_comm.p, ctx->d_comm.p, 2, NCCL_INT64, NCCL_MIN, ctx->comm, ctx->stream);
    if (r != 0)
        return ctx->fail(LVBGPU_E_COMM,
                         std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"));
    HIPCHK(ctx, hipMemcpyAsync(vals, ctx->d_comm.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *value = vals[0];
    if (argmin_rank)
        *argmin_rank = (int32_t)(vals[1] & 0xFFFF);
    return LVBGPU_OK;
}

extern "C" int lvbgpu_comm_destroy(lvbgpu_ctx *ctx)
{
    if (!ctx)
        return LVBGPU_E_ARG;
    if (ctx->comm && g_rccl.CommDestroy)
        (void)g_rccl.CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    return LVBGPU_OK;
}
