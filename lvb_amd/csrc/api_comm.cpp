// api_comm.cpp - liblvbgpu.so: the one collective of the path, a min-reduce of the best length over RCCL (loaded at run time).
#include "ctx.hpp"

namespace lvbgpu_detail
{
static Rccl g_rccl;
} // namespace lvbgpu_detail

// =================================================================================== RCCL

namespace lvbgpu_detail
{
bool load_rccl(std::string *why)
{
    if (g_rccl.lib)
        return true;
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    // a copy the process already holds (torch brings its own) first: one RCCL per process
    for (const char *nm : names)
        if ((g_rccl.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD)))
            break;
    for (const char *nm : names)
        if (!g_rccl.lib)
            g_rccl.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
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
constexpr int NCCL_SUM = 0;   // ncclSum
} // namespace lvbgpu_detail

extern "C" int lvbgpu_comm_available(void)
{
    std::string why;
    if (load_rccl(&why))
        return LVBGPU_OK;
    g_last_error_noctx = why;
    return LVBGPU_E_COMM;
}

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
    HIPCHK(ctx, ctx->d_comm.reserve(4096));
    return LVBGPU_OK;
}

extern "C" int lvbgpu_comm_size(const lvbgpu_ctx *ctx) { return ctx && ctx->comm ? ctx->comm_size : 1; }

extern "C" int lvbgpu_allreduce_min(lvbgpu_ctx *ctx, int64_t *value, int32_t *argmin_rank)
{
    if (!ctx || !value)
        return LVBGPU_E_ARG;
    if (!ctx->comm)
        return ctx->fail(LVBGPU_E_STATE, "no communicator: call lvbgpu_comm_init first");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // one 8-byte min over xGMI finds the best length; a second one over (length, rank) keys
    // names a rank that holds it.  Lengths are < 2^47 (MAX_M * 2 * MAX_N), ranks < 2^16.
    // A rank without a length yet (INT64_MAX, anything from 2^47 on) takes part with the largest key: the key is built
    // in unsigned arithmetic from a clamped value, so nothing overflows.
    constexpr int64_t NO_LENGTH = (int64_t)1 << 47;
    const int64_t mine = *value >= NO_LENGTH || *value < 0 ? NO_LENGTH : *value;
    long long vals[2] = {(long long)mine, (long long)(((uint64_t)mine << 16) | (uint64_t)(ctx->comm_rank & 0xFFFF))};
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_comm.p, vals, 16, hipMemcpyHostToDevice, ctx->stream));
    const int r = g_rccl.AllReduce(ctx->d_comm.p, ctx->d_comm.p, 2, NCCL_INT64, NCCL_MIN, ctx->comm, ctx->stream);
    if (r != 0)
        return ctx->fail(LVBGPU_E_COMM,
                         std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"));
    HIPCHK(ctx, hipMemcpyAsync(vals, ctx->d_comm.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *value = vals[0] >= NO_LENGTH ? INT64_MAX : vals[0];
    if (argmin_rank)
        *argmin_rank = vals[0] >= NO_LENGTH ? -1 : (int32_t)(vals[1] & 0xFFFF);
    return LVBGPU_OK;
}

// The second way the path shards (the reference's OpenMP slices, TreeEvaluation.c:95-97, 166-179): the
// SITE axis.  A length is a sum over sites, so when rank r's context holds columns [m r / k, m (r + 1) / k) of the
// alignment, a candidate's length is the sum of the ranks' lengths for it: one RCCL sum over B 64-bit values per step.
extern "C" int lvbgpu_allreduce_sum(lvbgpu_ctx *ctx, int64_t *values, int32_t count)
{
    if (!ctx || !values || count < 1)
        return LVBGPU_E_ARG;
    if (!ctx->comm)
        return ctx->fail(LVBGPU_E_STATE, "no communicator: call lvbgpu_comm_init first");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, ctx->d_comm.reserve((size_t)count * 8));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_comm.p, values, (size_t)count * 8, hipMemcpyHostToDevice, ctx->stream));
    const int r = g_rccl.AllReduce(ctx->d_comm.p, ctx->d_comm.p, (size_t)count, NCCL_INT64, NCCL_SUM, ctx->comm, ctx->stream);
    if (r != 0)
        return ctx->fail(LVBGPU_E_COMM,
                         std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"));
    HIPCHK(ctx, hipMemcpyAsync(values, ctx->d_comm.p, (size_t)count * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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

