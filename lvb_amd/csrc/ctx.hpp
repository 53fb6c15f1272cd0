// ctx.hpp - what the api_*.cpp files of liblvbgpu.so share: buffers, the context and batch objects,
// the error macro, and the few helpers defined in api_core.cpp that the other files call.
#pragma once

#include "../../include/lvbgpu.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
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

namespace lvbgpu_detail
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
} // namespace lvbgpu_detail

using namespace lvbgpu_detail;

// what belongs to ONE resident tree (a chain).  The selected chain's copy lives in the context's own fields
// (topo, topo_version, have_tree, ...); the others wait in lvbgpu_ctx::parked until lvbgpu_select_chain swaps them in.
struct ChainSlot
{
    Topology topo;
    uint64_t topo_version = 0;
    bool have_tree = false;
    int64_t cur_length = 0;
    bool cur_length_stale = false;
    uint64_t d_topo_version = ~0ull;
    uint32_t gen_table_bytes = 0;
    int32_t gen_K = 1;
};

// Host-side waits for the device are bounded: a spin on a flag, an event or a stream gives up after the context's wait
// limit (lvbgpu_set_wait_limit; default LVBGPU_WAIT_SECONDS or 30 s) and the call returns LVBGPU_E_HIP naming what it
// waited for and the stream's state - a stuck stream must not become an endless host spin.
struct WaitClock
{
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double limit;
    explicit WaitClock(double seconds) : limit(seconds) {}
    double waited() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
    bool expired() const { return waited() > limit; }
};

struct lvbgpu_ctx
{
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr; // read-backs of device-built batches' lengths: beside the next batch, not before it
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    long n = 0, nwords = 0;
    int32_t nb = 0;
    uint32_t stride_words = 0, stride4 = 0, ntiles = 0;
    uint32_t target_waves = TARGET_WAVES; // tuning knob (env LVBGPU_TARGET_WAVES)

    // several resident trees (chains) in one context: they share the leaf rows, chain c's internal node v (>= n)
    // has its row and its change slot at v + c (n - 3); lvbgpu_set_chains sizes the buffers, lvbgpu_select_chain
    // says which tree the single-tree calls mean (chain 0 by default)
    int32_t nchains = 1, chain = 0;
    std::vector<ChainSlot> parked;           // [nchains]; entry `chain` is stale while that chain is selected
    uint64_t version_counter = 0;            // topology versions are unique across chains (workers key copies on them)
    uint32_t rows_total() const { return (uint32_t)(n + (long)nchains * (n - 3)); }
    uint32_t chain_rows() const { return (uint32_t)(n - 3); }
    uint32_t row_of(int32_t v) const { return v < n ? (uint32_t)v : (uint32_t)(v + (long)chain * (n - 3)); }
    uint32_t root_slot() const { return rows_total() + (uint32_t)chain; }
    long long *scalars() const { return d_scalars + 4 * (size_t)chain; }
    uint64_t version_of(int32_t c) const { return c == chain ? topo_version : parked[(size_t)c].topo_version; }

    uint64_t *d_rows = nullptr;              // [rows_total][stride_words]
    unsigned long long *d_changes = nullptr; // [rows_total + nchains]; the last nchains slots = the chains' two root combines
    long long *d_scalars = nullptr;          // per chain 4: [0] S_all, [1] current length; [2] of chain 0: finished-wave count
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
    static constexpr int STEP_PIPELINE = 4;
    lvbgpu_batch *step_batch[STEP_PIPELINE] = {nullptr, nullptr, nullptr, nullptr}; // recycled by lvbgpu_score_batch
    bool pipeline_steps = true; // env LVBGPU_PIPELINE=0: one build, one launch per lvbgpu_score_batch
    lvbgpu_batch *full_batch = nullptr; // recycled by lvbgpu_score_full_batch
    // device-side proposals (lvbgpu_propose_score)
    // device-built batches: two may be in flight (lvbgpu_chains_submit / _collect), the plain calls use slot 0
    struct PSeg
    {
        int32_t chain, start, count;
        uint64_t version; // that chain's tree when the candidates were drawn
    };
    struct PropSlot
    {
        lvbgpu_batch *batch = nullptr;
        DevBuf d_pedits, d_pinfo;   // the candidates' rewrites and descriptors
        std::vector<PSeg> segs;     // the segments of the batch (lvbgpu_chains_commit picks from them)
        int32_t p_B = 0;            // candidates lvbgpu_proposal_edits may name (single chain, selected; 0: none)
        int32_t B = 0;
        hipEvent_t done_ev = nullptr; // after the lengths' read-back (on the copy stream)
        hipEvent_t walk_ev = nullptr; // after the walk (on the main stream): what the read-back waits for
        uint64_t submit_ord = 0;      // this batch's ordinal among the context's submits (1-based)
        PinBuf h_flag;                // the watcher's word for this slot's batches (kernels.hpp WalkArgs::watcher)
        uint32_t seq = 0;
        bool watched = false;         // this batch's lengths come through the watcher, not a copy
        bool in_flight = false;
        // a STEP (lvbgpu_chains_step_submit): the accept decision rides with the batch
        bool step = false;            // rules were given: the library decides at the collect and commits the picks
        std::vector<DecideRule> host_rules; // by draw index
    };
    static constexpr int PROP_SLOTS = 2;
    PropSlot pslot[PROP_SLOTS];
    int last_slot = 0;              // the batch lvbgpu_chains_commit picks from: the one collected last
    uint64_t submits = 0;           // device-built batches submitted so far
    uint64_t collected_ord = 0;     // highest submit ordinal among the batches collected so far
    double wait_limit_s = 30.0;     // lvbgpu_set_wait_limit
    uint32_t flip_counter = 0;      // direction of this context's next big scoring launch (launch_walk)
    DevBuf d_gen_prof; // LVBGPU_GEN_PROFILE: clock stamps of the generator (diagnostic)
    DevBuf d_post_prof; // LVBGPU_POST_PROFILE: clock stamps of the last post launch's workgroups (diagnostic)
    DevBuf d_topo4; // the generator's tables of the resident topologies, gen_table_stride each
    uint32_t gen_table_stride = 0;
    int32_t gen_kmax = 1; // ancestor tables hold 2^0 .. 2^(kmax-1): 2^kmax exceeds any depth of a tree of these taxa
    PinBuf h_pinfo, h_topo;
    std::vector<uint16_t> gen_tab16;
    std::vector<int32_t> gen_tab32;
    uint32_t gen_table_bytes = 0, gen_idx_bytes = 2;
    int32_t gen_K = 1;
    DevBuf d_moves; // moves named by the host (lvbgpu_score_moves)
    PinBuf h_moves;
    uint64_t d_topo_version = ~0ull;
    uint32_t p_stride_t = 0, p_stride_e = 0;
    // lvbgpu_chains_commit: picks / fetched rewrites travel through a small ring of pinned slots
    static constexpr int PICK_SLOTS = 4;
    PinBuf h_pick[PICK_SLOTS];
    // A slot's readers (commit walk on the main stream, table rebuild and gather on the side stream) are finished once
    // a batch SUBMITTED AFTER that use has been collected: its walk was enqueued behind the commit walk, its generator
    // waited for the rebuild.  Every use remembers how many batches had been submitted before it (api_propose.cpp
    // take_pick_slot); a slot is taken again only when a later batch has come back, else the streams are drained first.
    uint64_t pick_use_ord[PICK_SLOTS] = {0, 0, 0, 0};
    bool pick_used[PICK_SLOTS] = {false, false, false, false};
    int pick_slot = 0;
    uint32_t pick_seq = 0;
    int last_pick_count = 0;                     // picks of the last lvbgpu_chains_commit
    bool last_pick_has[MAX_CHAINS] = {};         // ... and which of their records exist
    std::vector<char> pick_records;              // ... copied out of the pinned slot when they arrived (lvbgpu_chains_picked_edits)
    uint32_t pick_records_stride = 0;
    std::vector<int32_t> step_map;               // ... its draws -> record index of lvbgpu_chains_picked_edits (-1: nothing accepted)
    DevBuf d_done; // per picked candidate: finished-wave count of a multi-chain commit (zero between launches)
    // The host side of the last lvbgpu_chains_commit, not done yet: the picked moves' descriptors and rewrites are on
    // their way into pinned slot `slot` (flag = seq), and the chains' host topologies follow when somebody needs them
    // (resolve_follow / settle in api_propose.cpp) - the commit call itself does not wait: on the accept path the host
    // is what the device waits for, and what the host has to do next (plan and submit the next step) needs no topology.
    // Versions are bumped at the commit already; ChainSlot::topo of the listed chains is one move behind until then.
    struct Follow
    {
        bool pending = false;
        int slot = 0;
        int32_t k = 0;
        uint32_t seq = 0;
        int32_t chains[MAX_CHAINS];
        bool has[MAX_CHAINS]; // record j exists
    } follow;
    // What lvbgpu_chains_commit, lvbgpu_chains_reroot and lvbgpu_chains_commit_edits have been asked for but the device
    // has not been given yet: commits of several chains are collected here and go out as ONE post launch (kernels.hpp
    // PostArgs: commit walk + table rebuilds + the moves' records to the host) - together with the next step's generator
    // when the next thing on this context is a submit, which in an annealing loop it is (api_propose.cpp flush_pending;
    // every other entry point flushes first, so nobody sees a tree the device has not caught up with).
    struct Pending
    {
        // accepted candidates of the device-built batch in pslot[src_slot]: launch candidates 0 .. k_pick - 1
        int32_t k_pick = 0;
        int src_slot = 0;
        uint32_t where[MAX_CHAINS];
        int32_t pick_chain[MAX_CHAINS];
        bool pick_rebuild = false;  // their chains' tables follow on the device
        int gather_slot = 0;        // pinned slot their records go to (lvbgpu_ctx::follow waits for gather_seq there)
        uint32_t gather_seq = 0;
        // moves named by the host (re-roots, host-made candidates), programs and rewrites in pinned slot ext_slot:
        // launch candidates k_pick .. k_pick + k_ext - 1
        int32_t k_ext = 0;
        int ext_slot = 0;
        size_t ext_o_t = 0, ext_o_d = 0, ext_o_x = 0, ext_o_e = 0;
        int32_t ext_max_stack = 1;
        bool ext_rebuild = false;
        uint64_t chains = 0; // every chain above, as a mask: one move per chain and launch
        bool any() const { return k_pick > 0 || k_ext > 0; }
    } pend;
    DevBuf d_table_ready; // uint32[MAX_CHAINS]: = post_seq once a post launch has rebuilt that chain's tables
    uint32_t post_seq = 0;
    int64_t post_launches = 0, post_launches_with_generator = 0; // (lvbgpu_debug_count)
    PinBuf h_pin;
    // direct steps: small batches whose programs the walk reads straight from h_pin and whose lengths its last
    // wave writes straight into the batch's pinned buffer; the host polls h_step's first word for step_seq
    PinBuf h_step;
    uint32_t step_seq = 0;
    bool direct_steps = true; // env LVBGPU_DIRECT_STEPS=0 turns them off (A/B measurements)
    bool starve_watcher = false; // env LVBGPU_DEBUG_STARVE_WATCHER (test hook, kernels.hpp WalkArgs::watch_starve)
    bool lpt_order = true;    // env LVBGPU_LPT=0: keep big batches in the caller's order on the device
    // the candidates of the last lvbgpu_chains_score_edits call: lvbgpu_chains_commit_edits walks an accepted one's SCORED
    // program (it lies in the batch) instead of building it again
    struct ScoredEdit
    {
        int32_t chain, n_edits, edit_off; // its rewrites: scored_edits[edit_off .. edit_off + n_edits)
        uint64_t version, hash;
    };
    std::vector<ScoredEdit> scored;
    std::vector<lvbgpu_edit> scored_edits;
    lvbgpu_batch *scored_batch = nullptr;
    uint64_t scored_gen = 0;
    int64_t commits_reusing_programs = 0;
    int64_t paired_walks = 0; // scoring walks launched two candidates per wave (lvbgpu_debug_paired_walks)
    int pair_min = 0;         // env LVBGPU_PAIR=n: batches of n candidates and more are walked two candidates per wave
    DevBuf d_tmp_changes;     // fused commits: per-combine accumulators, zero between launches
    size_t tmp_changes_zeroed_cap = 0; // capacity of d_tmp_changes when it was last cleared (0: never)
    DevBuf d_cin, d_cout; // strict-compat arenas
    PinBuf h_cin, h_cout;
    std::vector<int32_t> slot_of;
    std::vector<uint32_t> slot_epoch;
    uint32_t slot_gen = 0;

    void *comm = nullptr;
    int comm_rank = 0, comm_size = 1;
    DevBuf d_comm;

    // batches the caller holds (lvbgpu_batch_build): lvbgpu_destroy detaches them, so that a batch freed
    // after its context touches nothing of it
    std::vector<lvbgpu_batch *> held;
    // a bounded wait of lvbgpu_score_batch gave up: its recycled step batches may still be read and written by queued
    // walks, so they are set aside (freed with the context) instead of being reused or drained for
    bool wait_gave_up = false;
    std::vector<lvbgpu_batch *> set_aside;

    // lvbgpu_walk_timing: HIP events around every scoring walk, on the stream it is launched on
    static constexpr int WT_RING = 32;
    bool walk_timing = false;
    uint32_t wt_every = 1;  // every wt_every-th walk is timed (events between launches cost ~10 us each)
    uint64_t wt_seen = 0;
    hipEvent_t wt_ev[2 * WT_RING] = {};
    int wt_pending = 0;
    double wt_ms = 0.0;
    int64_t wt_launches = 0;
    DevBuf d_probe_sink;

    std::string last_error;

    int fail_hip(hipError_t e, const char *what)
    {
        last_error = std::string(what) + ": " + hipGetErrorString(e);
        if (e == hipErrorNotReady) // only a bounded wait that gave up returns this
            last_error += " (the wait limit of " + std::to_string(wait_limit_s) + " s passed: lvbgpu_set_wait_limit / LVBGPU_WAIT_SECONDS)";
        return (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver)
                   ? LVBGPU_E_NODEVICE
                   : (e == hipErrorOutOfMemory ? LVBGPU_E_NOMEM : LVBGPU_E_HIP);
    }
    // a bounded host wait gave up (WaitClock)
    int fail_wait(const char *what, double seconds)
    {
        char buf[64];
        snprintf(buf, sizeof buf, "%.3f", seconds);
        last_error = std::string(what) + " did not complete within " + buf +
                     " s (wait limit: lvbgpu_set_wait_limit / LVBGPU_WAIT_SECONDS); the stream is still busy or stuck";
        return LVBGPU_E_HIP;
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
    PinBuf h_stage; // recycled batches: the programs as the host built them (copied from here, or read in place)
    size_t off_toks = 0, off_dsts = 0;
    lvbgpu_batch_stats stats{};
    bool full_mode = false; // whole topologies: reads leaf rows only
    // step batches owned by the context (lvbgpu_score_batch / _score_full_batch) run build -> launch
    // -> lengths back to back, which lets them drop one synchronisation and take the zeroing of the
    // length slots off the critical path
    bool recycled = false;
    bool len_zeroed = false; // d_len was cleared after the previous read-back
    bool direct = false;     // this step's lengths come back through the walk's last wave (no copy, no stream wait)
    bool in_place = false;   // ... and its programs are read where they lie in ctx->h_pin
    bool launched = false;   // lengths exist (or are on their way)
    // device-built batches: the walk's watcher waves store the lengths into h_len and set watch_flag[0 .. WATCH_WAVES)
    // to watch_seq (pinned; null: no watcher, the caller copies the lengths back)
    uint32_t *watch_flag = nullptr;
    uint32_t watch_seq = 0;
    // host-built step batches (recycled, too large for a direct step): watcher waves of their own, so that the lengths need
    // no read-back copy and no stream synchronisation behind the walk
    PinBuf h_wflag;
    uint32_t own_seq = 0;
    bool own_watch = false;
    bool spans_chains = false; // device-built batch over several chains: every program names its own chain
    uint64_t build_gen = 0;    // counts the builds into this batch (lvbgpu_chains_commit_edits re-uses scored programs)
    uint64_t topo_version = 0; // resident tree the programs were built against (edits are relative to it)
    int32_t chain = 0;         // ... and which chain's tree that is
    std::vector<int32_t> slot_of; // big batches: candidate b sits at position slot_of[b] (longest program first)
    // two candidates per wave (kernels.hpp WalkArgs::pairs): who walks with whom, npairs == 0: one candidate per wave
    DevBuf d_pairs;
    PinBuf h_pairs;
    uint32_t npairs = 0;
};

// steps up to this many candidates finish within a few hundred microseconds: poll for them instead of
// sleeping in the runtime (its wake-up costs ~10 us per step)
constexpr int32_t SPIN_WAIT_MAX_B = 16384;
// direct steps: lengths come back through the walk's last wave while the launch has at most this many waves
// (every wave ticks one counter: measured on MI355X a gain up to a few hundred waves, a loss from ~1000) ...
constexpr uint32_t DIRECT_STEP_MAX_ITEMS = 512;
// ... and programs are read in place (pinned host memory) while all the waves together fetch at most this much
// over the host link (measured: faster than the copy up to ~64 KiB, slower beyond)
constexpr size_t DIRECT_READ_MAX_BYTES = 64u << 10;
constexpr int32_t STEP_PIPELINE_PIECE = 2048; // lvbgpu_score_batch cuts a batch into up to 4 pieces of at least this size
constexpr int32_t LPT_MIN_B = 2048; // from here on a launch is many rounds of waves and its tail shows

// the first statement of every entry point that enqueues device work or reads device state: select the device and let
// the device catch up with the commits collected in lvbgpu_ctx::pend
#define ENTER(ctx)                                                                                                     \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t es__ = hipSetDevice((ctx)->device);                                                                 \
        if (es__ != hipSuccess)                                                                                        \
            return (ctx)->fail_hip(es__, "hipSetDevice");                                                              \
        if ((ctx)->pend.any())                                                                                         \
        {                                                                                                              \
            const int ef__ = flush_pending((ctx), nullptr);                                                            \
            if (ef__ != LVBGPU_OK)                                                                                     \
                return ef__;                                                                                           \
        }                                                                                                              \
    } while (0)

#define HIPCHK(ctx, call)                                                                                              \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e__ = (call);                                                                                       \
        if (e__ != hipSuccess)                                                                                         \
            return (ctx)->fail_hip(e__, #call);                                                                        \
    } while (0)


namespace lvbgpu_detail
{
extern thread_local std::string g_last_error_noctx; // last error of calls that have no context yet

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

inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

// api_core.cpp
int hip_status_noctx(hipError_t e, const char *what);
hipError_t wait_for_step(lvbgpu_ctx *ctx, int32_t B);
WalkArgs resident_args(lvbgpu_ctx *ctx, const void *prog, size_t off_toks, size_t off_dsts, void *d_len, uint32_t B,
                       int32_t max_stack);
int check_depth(lvbgpu_ctx *ctx, int32_t max_stack);
int read_current_length(lvbgpu_ctx *ctx);
int run_commit_program(lvbgpu_ctx *ctx, const Program &prog, bool zero_all, bool readback);
int walk_timing_drain(lvbgpu_ctx *ctx);
// api_propose.cpp: finish the host side of the last lvbgpu_chains_commit (Follow).  resolve_follow: every chain parked
// (inside a multi-chain call); settle: from anywhere - every entry point that reads a chain's host topology calls it
int resolve_follow(lvbgpu_ctx *ctx);
uint64_t edits_hash(const lvbgpu_edit *e, int32_t n);
int settle(lvbgpu_ctx *ctx);
// api_propose.cpp: give the device what lvbgpu_ctx::pend holds (one post launch; with `gen` the next batch's generator rides
// in it).  Nothing pending and no generator: nothing happens.
int flush_pending(lvbgpu_ctx *ctx, const GenArgs *gen);
} // namespace lvbgpu_detail

using namespace lvbgpu_detail;
