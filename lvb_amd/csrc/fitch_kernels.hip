// fitch_kernels.hip - hand-written gfx950 (MI355X, CDNA4) kernels of the Fitch scoring path.
//
// What is computed is exactly getplen's arithmetic (reference TreeEvaluation.c:204-264): for a
// list of dirty nodes in postorder, z = fitch(x, y) per site with the number of sites that needed
// a union counted, plus the two root combines.  How it is laid out is native:
//
//   * BIT-PLANE state sets in HBM.  The reference packs one 4-bit set per nibble (LVB.h:72-89)
//     and needs ~9 ALU ops per 8 sites to find empty intersections inside nibbles.  On the
//     device each 16-byte group of a row holds the SAME 32 sites as the reference's two 64-bit
//     words, but transposed: four 32-bit planes (A, C, G, T), bit s = site s.  The Fitch step is
//     then 9 VALU ops per 32 sites: three v_bitop3 build any = OR_b(x_b & y_b), one v_bcnt
//     accumulates the non-empty count, four v_bitop3 select (any ? x&y : x|y) per plane.
//     The transposition is a per-lane bit shuffle applied once on the way in (leaf rows, strict-
//     compat operands) and on the way out (lvbgpu_get_sets, strict-compat results); lengths and
//     change counts do not depend on it.
//   * one wavefront (64 lanes) walks one candidate's postorder program for one 2048-site tile:
//     a lane owns one 16-byte group, so every row load is one coalesced 1 KiB
//     global_load_dwordx4 per wave;
//   * the program's tokens are fetched once per 64 with a coalesced vector load (lane k holds
//     token k and its row offset) and handed to the scalar unit with v_readlane; row loads run
//     four tokens ahead of the arithmetic in a 4-slot register ring (they depend only on the
//     program, never on computed values), waited for with counted vmcnt;
//   * the running node set lives in 4 VGPRs; sibling sets that must wait for the other subtree sit
//     on an operand stack in LDS (ds_write_b128 / ds_read_b128, 1 KiB per level per wave: two
//     levels for NNI/SPR/TBR deltas, up to log2(n)+1 for whole-tree programs);
//   * union counts are reduced across the wave (a butterfly over the LDS crossbar for a program's total,
//     ballot + scalar popcount bit-slices for a commit's per-node counts) and added with one integer atomic
//     per wave: integer addition commutes, so lengths are bit-exact whatever the arrival order;
//   * the commit form parks produced sets and counts in LDS and writes them out in bursts, so that the load
//     ring's counted waits survive (a write in flight would turn each of them into vmcnt(0));
//   * small launches hand their results to the host themselves: the last wave to finish (a counter every wave
//     ticks) copies lengths into pinned memory and releases a flag (direct steps), or settles changes[] and
//     S_all after a commit (fused commits);
//   * blockIdx -> work mapping is XCD-aware: items are ordered tile-major and each of the 8 XCDs
//     takes a contiguous eighth, so the waves resident on one XCD read the same column slice of
//     the resident tree and hit that XCD's private 4 MiB L2 (measured hit rate 97 %).
//
// No MFMA: bitwise integer streaming, bound by the L2 -> CU path.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <cstdlib>
#include <stdint.h>

#include "kernels.hpp"
#include "walk_body.hpp"

namespace lvbgpu
{
// nibble layout (reference: nibble k of the 128-bit group = site k, bit0=A..bit3=T) <-> planes
__device__ __forceinline__ uint32_t gather_bit(uint32_t w, uint32_t b)
{
    uint32_t x = (w >> b) & 0x11111111u; // bit b of each of 8 nibbles, at positions 0,4,..,28
    x = (x | (x >> 3)) & 0x03030303u;
    x = (x | (x >> 6)) & 0x000F000Fu;
    x = (x | (x >> 12)) & 0xFFu;
    return x; // 8 sites
}
__device__ __forceinline__ uint32_t scatter_bits(uint32_t byte, uint32_t b)
{
    uint32_t x = byte & 0xFFu;
    x = (x | (x << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return x << b;
}
__device__ __forceinline__ uint4 nibbles_to_planes(const uint4 w)
{
    uint32_t p[4];
#pragma unroll
    for (uint32_t b = 0; b < 4; b++)
        p[b] = gather_bit(w.x, b) | (gather_bit(w.y, b) << 8) | (gather_bit(w.z, b) << 16) | (gather_bit(w.w, b) << 24);
    return make_uint4(p[0], p[1], p[2], p[3]);
}
__device__ __forceinline__ uint4 planes_to_nibbles(const uint4 p)
{
    uint32_t w[4];
#pragma unroll
    for (uint32_t q = 0; q < 4; q++)
        w[q] = scatter_bits(p.x >> (8 * q), 0) | scatter_bits(p.y >> (8 * q), 1) | scatter_bits(p.z >> (8 * q), 2) |
               scatter_bits(p.w >> (8 * q), 3);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// The watcher waves of a launch (WalkArgs::watcher; eight extra blocks at the end of the grid, wave `wid` of WATCH_WAVES):
// wait until every length slot of this wave's chunks has its ngroups arrivals and hand the lengths to the host.  Shared
// by fitch_walk and fitch_walk_pair.
__device__ __forceinline__ void watcher_block(const WalkArgs &a, const uint32_t wid, const uint32_t lane)
{
    // all 32 waves of the eight extra blocks watch: wave w takes the 64-candidate chunks w, w + 32, ... (one
    // wave alone would take B / 64 dependent round trips AFTER the last walking wave: 35 us at B = 4096)
    constexpr uint32_t WATCHERS = WATCH_WAVES;
    const unsigned long long want = (unsigned long long)a.ngroups + (a.watch_starve ? 1u : 0u);
    const unsigned long long count_mask = 0xFFFull << WATCH_COUNT_SHIFT;
    bool gave_up = false;
    for (uint32_t base = wid * 64u; base < a.B; base += WATCHERS * 64u)
    {
        const uint32_t i = base + lane;
        if (i < a.B)
        {
            unsigned long long v;
            uint32_t budget = a.watch_starve ? 4096u : 1u << 24; // ~ seconds: every walking wave finishes on its own, this is a backstop
            while ((((v = __hip_atomic_load(a.len_out + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & count_mask) >>
                    WATCH_COUNT_SHIFT) != want &&
                   --budget)
                __builtin_amdgcn_s_sleep(8);
            gave_up |= budget == 0u;
            __hip_atomic_store(a.host_len + i, v & ~count_mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    atomics_acknowledged(); // the wave's stores (all lanes) before its flag
    // every watcher wave has a flag word of its own (the host waits for all 32): no counter for them to meet at
    if (lane == 0)
        __hip_atomic_store(a.host_flag + wid, __builtin_amdgcn_ballot_w64(gave_up) != 0ull ? 0xFFFFFFFFu : a.step_seq,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// WIDE: row offsets kept as 64-bit byte counts in two vectors (tree blocks of 64 GiB and more).  The
// narrow form - offsets in 16-byte units, 32 bits, the shift folded into the address add - saves three
// instructions per token: 2.3 % of the launch (the loop sits within 7 % of what the L2 -> CU path delivers,
// instruction count is the second-order term: DESIGN.md section 3).
// (Two items per wave with both descriptors and token vectors requested up front was tried and measured 2-5 % slower:
// profiles/experiments/r02_walk_and_step.md; commit 3b695f1 still has it.)
// (HANDOVER, below: with the watcher as a run-time branch of the one kernel the walk took 92.8 instead of 87.5 us)
template <bool COMMIT, bool WIDE, int HANDOVER = 0>
__global__ __launch_bounds__(WALK_THREADS) __attribute__((amdgpu_waves_per_eu(COMMIT ? 4 : 8))) void fitch_walk(const WalkArgs a)
{
    extern __shared__ uint4 lds_stack[]; // operand stack: [wave][level][lane]

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // XCD-aware remap: hardware deals consecutive block ids round-robin over the 8 XCDs; give
    // XCD x the x-th contiguous eighth of the tile-major item list.  gridDim.x % 8 == 0.
    // (a watcher launch has eight more blocks than item blocks: the first of them watches, see WalkArgs::watcher)
    // HANDOVER (walk_body.hpp): kernels of their own, so that the plain walk carries none of the other two's code.
    constexpr bool WATCH = HANDOVER == 2;
    const uint32_t nblk = WATCH ? gridDim.x - 8u : gridDim.x;
    if constexpr (WATCH)
        if (blockIdx.x >= nblk)
        {
            watcher_block(a, (blockIdx.x - nblk) * WALK_WAVES + wave, lane);
            return;
        }
    // a.flip: every other launch walks each XCD's share of the tile-major list from its far end.  A tree block beyond
    // the XCD L2s (cfg5: 401 MB) is then re-read starting with the tiles the previous launch touched LAST, which are
    // the ones still in the 256 MiB Infinity Cache (walking the same way every time would find none of it there).
    const uint32_t in_xcd = a.flip ? (nblk >> 3) - 1u - (blockIdx.x >> 3) : (blockIdx.x >> 3);
    const uint32_t pos = (blockIdx.x & 7u) * (nblk >> 3) + in_xcd;
    const uint32_t item = pos * WALK_WAVES + wave;
    if (item >= a.nitems)
        return;
    walk_item<COMMIT, WIDE, HANDOVER>(a, lds_stack, lane, wave, WALK_WAVES, item);
}

// ---------------------------------------------------------------------------------------------
// Two candidates per wave (scoring launches; WalkArgs::pairs).  Candidates whose programs END alike - the same clean rows
// with the same flags from some token on to the root: moves whose dirty paths run together - are handed to ONE wave,
// which walks A's private part, then B's, then the common suffix ONCE with two states (accumulator, operand stack,
// counters): every row of the suffix is loaded once and combined twice.  The walk is bound by what the L2 -> CU path
// feeds a wave, and a pure-load probe with this shape - half the waves, as many loads fewer as
// the pairs share - takes proportionally less time (tools/pair_probe.py: 4096 x 28 loads 90.9 us, 2048 x 40 loads
// 64.7 us).  Who is paired with whom is the producer's business (host: sorted by the program read backwards; device:
// every generating workgroup among the sixteen candidates it has drawn); the wave finds the suffix itself by comparing the
// two token vectors from the end.  Programs longer than one 64-token chunk are walked alone (b = NONE, or no sharing).
struct WalkState
{
    uint4 acc;
    uint32_t nonempty, nonempty_rare, sp;
};

template <bool WIDE, bool WATCH>
__global__ __launch_bounds__(WALK_THREADS) void fitch_walk_pair(const WalkArgs a)
{
    extern __shared__ uint4 lds_stack[]; // operand stacks: [wave][state][level][lane]

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nblk = WATCH ? gridDim.x - 8u : gridDim.x;
    if constexpr (WATCH)
        if (blockIdx.x >= nblk)
        {
            watcher_block(a, (blockIdx.x - nblk) * WALK_WAVES + wave, lane);
            return;
        }
    const uint32_t in_xcd = a.flip ? (nblk >> 3) - 1u - (blockIdx.x >> 3) : (blockIdx.x >> 3);
    const uint32_t pos = (blockIdx.x & 7u) * (nblk >> 3) + in_xcd;
    const uint32_t item = pos * WALK_WAVES + wave;
    if (item >= a.nitems)
        return;
    // an item = (tile group, pair); inv_B was made for npairs by launch_walk
    uint32_t group = __umulhi(item, a.inv_B), pair = item - group * a.npairs;
    if (pair >= a.npairs)
    {
        group++;
        pair -= a.npairs;
    }
    const uint32_t tile_begin = group * a.tiles_per + (group < a.tiles_rem ? group : a.tiles_rem);
    const uint32_t tile_end = tile_begin + a.tiles_per + (group < a.tiles_rem ? 1u : 0u);

    // (everything about the pair is wave-uniform and belongs in scalar registers: the indices come from memory, so the
    // compiler cannot know - told so explicitly; with both descriptors in vector registers the walk spilled)
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    auto uniform_desc = [&](uint32_t idx) {
        const CandDesc v = a.cands[idx];
        CandDesc u;
        u.tok_off = uni(v.tok_off);
        u.ntok = uni(v.ntok);
        u.dst_off = uni(v.dst_off);
        u.ncomb = uni(v.ncomb);
        u.base = (long long)(((unsigned long long)uni((uint32_t)((unsigned long long)v.base >> 32)) << 32) | uni((uint32_t)v.base));
        u.flags = uni(v.flags);
        u.nfresh = uni(v.nfresh);
        return u;
    };
    const uint32_t ia = uni(a.pairs[2u * pair]), ib = uni(a.pairs[2u * pair + 1u]);
    const bool two = ib != PICK_NONE;
    const CandDesc cdA = uniform_desc(ia);
    CandDesc cdB{};
    if (two)
        cdB = uniform_desc(ib);
    const uint32_t *__restrict__ tkA = a.toks + cdA.tok_off;
    const uint32_t *__restrict__ tkB = a.toks + cdB.tok_off;
    const uint32_t chainA = cdA.flags >> CAND_CHAIN_SHIFT, chainB = cdB.flags >> CAND_CHAIN_SHIFT;

    // Both token vectors are fetched NOW, together (programs within one chunk): a wave's time is its chain of dependent
    // round trips, and every phase below would otherwise begin with one of its own.
    const bool preA = cdA.ntok <= 64u, preB = cdB.ntok <= 64u;
    uint32_t tokA = 0u, tokB = 0u;
    if (preA && lane < cdA.ntok)
        tokA = tkA[lane];
    if (two && preB && lane < cdB.ntok)
        tokB = tkB[lane];
    // the common suffix: tokens compared from the end, one per lane (the vectors turned round over the LDS crossbar)
    uint32_t k = 0;
    if (two && preA && preB && cdA.ntok != 0u && cdB.ntok != 0u && chainA == chainB)
    {
        uint32_t ra_ = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((cdA.ntok - 1u - lane) & 63u) << 2), (int)tokA);
        uint32_t rb_ = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((cdB.ntok - 1u - lane) & 63u) << 2), (int)tokB);
        ra_ = lane < cdA.ntok ? ra_ : 0xFFFFFFFEu;
        rb_ = lane < cdB.ntok ? rb_ : 0xFFFFFFFDu;
        const uint64_t differ = ~__builtin_amdgcn_ballot_w64(ra_ == rb_);
        k = differ ? (uint32_t)__builtin_ctzll(differ) : 64u;
        const uint32_t shortest = cdA.ntok < cdB.ntok ? cdA.ntok : cdB.ntok;
        k = k < shortest ? k : shortest;
    }

    const char *__restrict__ in = reinterpret_cast<const char *>(a.rows_in) + (uint64_t)tile_begin * a.in_tile_bytes + lane * 16u;
    global_cp lane_ptr = (global_cp)in;
    auto load_row = [&](uint64_t off) -> uint4 {
        asm volatile("" : "+v"(lane_ptr));
        const u32x4 v = *reinterpret_cast<const __attribute__((address_space(1))) u32x4 *>(lane_ptr + off);
        return make_uint4(v.x, v.y, v.z, v.w);
    };
    auto load_row16 = [&](uint32_t off16) -> uint4 {
        asm volatile("" : "+v"(lane_ptr));
        const u32x4 v = *reinterpret_cast<const __attribute__((address_space(1))) u32x4 *>(lane_ptr + ((uint64_t)off16 << 4));
        return make_uint4(v.x, v.y, v.z, v.w);
    };
    uint4 *const stackA = lds_stack + (size_t)(wave * 2u) * a.stack_depth * 64u + lane;
    uint4 *const stackB = stackA + (size_t)a.stack_depth * 64u;
    const uint4 ones = make_uint4(~0u, ~0u, ~0u, ~0u);

    // what the lengths' bases need (the cached changes of the candidates' dirty nodes, S_all), requested with the token
    // vectors and reduced to wave-uniform values at once: nothing of it is carried through the walk in vector registers
    long long baseA = cdA.base, baseB = cdB.base;
    if (group == 0)
    {
        auto resident_base = [&](const CandDesc &cd, uint32_t chain) -> long long {
            if (!(cd.flags & CAND_RESIDENT_BASE))
                return 0;
            const int32_t *ds = a.dsts + cd.dst_off;
            const uint32_t bias = chain * a.chain_rows;
            long long sub = 0;
            for (uint32_t i = lane; i < cd.ncomb; i += 64u)
            {
                const int32_t dst = ds[i];
                if (dst >= 0)
                    sub += a.node_changes[(uint32_t)dst >= a.bias_from ? (uint32_t)dst + bias : (uint32_t)dst];
            }
            for (int off = 32; off > 0; off >>= 1)
                sub += __shfl_xor(sub, off);
            const long long s_all = a.s_all[4u * chain];
            const long long v = s_all - sub;
            return (long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)v));
        };
        baseA += resident_base(cdA, chainA);
        if (two)
            baseB += resident_base(cdB, chainB);
    }

    // tokens [lo, hi) of the program at `tk` (ntok tokens), for one state (S2 == nullptr-like: DUAL false) or for two that
    // share every load (DUAL true: the tokens are the same for both; lo .. hi lie within one chunk then)
    auto run = [&](auto dual_tag, const uint32_t *__restrict__ tk, const uint32_t ntok, const uint32_t lo, const uint32_t hi,
                   const uint32_t row_bias, WalkState &P, uint4 *const stP, WalkState &Q, uint4 *const stQ, const bool have_tok,
                   const uint32_t the_tok) __attribute__((always_inline)) {
        constexpr bool DUAL = decltype(dual_tag)::value;
        for (uint32_t c0 = lo & ~63u; c0 < hi; c0 += 64u)
        {
            const uint32_t cnt = (ntok - c0 < 64u) ? ntok - c0 : 64u;
            uint32_t mytok = have_tok ? the_tok : ((lane < cnt) ? tk[c0 + lane] : 0u);
            // (opaque: what follows - three lane-shifted offset vectors per phase - is made HERE, per phase and tile, not
            // hoisted out of the tile loop and kept alive through the other phases: the walk spilled its row pointer)
            asm volatile("" : "+v"(mytok));
            const uint32_t r0 = mytok & TOK_ROW_MASK;
            const uint32_t myrow = r0 >= a.bias_from ? r0 + row_bias : r0;
            const uint64_t myoff64 = WIDE ? (uint64_t)myrow * ((uint64_t)a.in_stride4 << 4) : (uint64_t)(myrow * a.in_stride4);
            OffVec o0{(uint32_t)myoff64, (uint32_t)(myoff64 >> 32)};
            auto down = [&](const OffVec &v, uint32_t d) {
                const int sel = (int)(((lane + d) & 63u) << 2);
                return OffVec{(uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)v.lo),
                              WIDE ? (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)v.hi) : 0u};
            };
            const OffVec o1 = down(o0, 1u), o2 = down(o0, 2u), o3 = down(o0, 3u);
            auto row_at = [&](const OffVec &v, uint32_t j) -> uint4 {
                const uint32_t lo32 = (uint32_t)__builtin_amdgcn_readlane((int)v.lo, (int)j);
                if constexpr (!WIDE)
                    return load_row16(lo32);
                const uint32_t hi32 = (uint32_t)__builtin_amdgcn_readlane((int)v.hi, (int)j);
                return load_row(((uint64_t)hi32 << 32) | lo32);
            };
            const uint64_t freshm = __builtin_amdgcn_ballot_w64((mytok & TOK_FRESH) != 0u);
            const uint64_t mergem = __builtin_amdgcn_ballot_w64(((mytok >> TOK_MERGE_SHIFT) & TOK_MERGE_MASK) != 0u);
            const uint64_t postm = mergem | (freshm >> 1);
            auto tok_at = [&](uint32_t j) { return (uint32_t)__builtin_amdgcn_readlane((int)mytok, (int)j); };
            auto post = [&](uint32_t j) __attribute__((always_inline)) {
                const uint32_t tok = tok_at(j);
                for (uint32_t m = (tok >> TOK_MERGE_SHIFT) & TOK_MERGE_MASK; m != 0; m--)
                {
                    P.sp--;
                    P.acc = fitch_planes(stP[(size_t)P.sp * 64u], P.acc, P.nonempty_rare);
                    if constexpr (DUAL)
                    {
                        Q.sp--;
                        Q.acc = fitch_planes(stQ[(size_t)Q.sp * 64u], Q.acc, Q.nonempty_rare);
                    }
                }
                if (j < 63u && ((freshm >> (j + 1u)) & 1u)) // the next token (of this chunk) starts a chain
                {
                    if (tok_at(j + 1u) & TOK_PUSH)
                    {
                        stP[(size_t)P.sp * 64u] = P.acc;
                        P.sp++;
                        if constexpr (DUAL)
                        {
                            stQ[(size_t)Q.sp * 64u] = Q.acc;
                            Q.sp++;
                        }
                    }
                    P.acc = ones;
                    if constexpr (DUAL)
                        Q.acc = ones;
                }
            };
            auto step = [&](uint32_t j, uint32_t flagged, const uint4 cur) __attribute__((always_inline)) {
                P.acc = fitch_planes(P.acc, cur, P.nonempty);
                if constexpr (DUAL)
                    Q.acc = fitch_planes(Q.acc, cur, Q.nonempty);
                if (__builtin_expect(flagged != 0u, 0))
                    post(j);
            };
            const uint32_t jb = lo > c0 ? lo - c0 : 0u;                  // first token of the range in this chunk
            const uint32_t je = hi - c0 < cnt ? hi - c0 : cnt;           // one past its last
            if (c0 != 0u && jb == 0u && (freshm & 1u)) // a chunk of a long program that opens with a chain start (one state only)
            {
                if (tok_at(0) & TOK_PUSH)
                {
                    stP[(size_t)P.sp * 64u] = P.acc;
                    P.sp++;
                }
                P.acc = ones;
            }
            const uint32_t n = je - jb;
            if (n < 4u)
            {
                for (uint32_t j = jb; j < je; j++)
                    step(j, (uint32_t)(postm >> j) & 1u, row_at(o0, j));
                continue;
            }
            // 4-slot ring: slot q holds the row of token j + q (as in fitch_walk, from token jb on)
            uint4 ra = row_at(o0, jb), rb = row_at(o1, jb), rc = row_at(o2, jb), rd = row_at(o3, jb);
            uint32_t jl = jb + 4u; // first token not yet in the ring
            uint64_t fm = postm >> jb;
#define LVB_PGROUP(F, J0, JL)                                                                                 \
    step((J0), (F) & 1u, ra);                                                                                 \
    ra = row_at(o0, (JL));                                                                                    \
    step((J0) + 1u, (F) & 2u, rb);                                                                            \
    rb = row_at(o1, (JL));                                                                                    \
    step((J0) + 2u, (F) & 4u, rc);                                                                            \
    rc = row_at(o2, (JL));                                                                                    \
    step((J0) + 3u, (F) & 8u, rd);                                                                            \
    rd = row_at(o3, (JL));
            for (; jl + 8u <= je; jl += 8u, fm >>= 8)
            {
                const uint32_t f = (uint32_t)fm;
                LVB_PGROUP(f, jl - 4u, jl)
                LVB_PGROUP(f >> 4, jl, jl + 4u)
            }
            if (jl + 4u <= je)
            {
                const uint32_t f = (uint32_t)fm;
                LVB_PGROUP(f, jl - 4u, jl)
                jl += 4u;
                fm >>= 4;
            }
#undef LVB_PGROUP
            const uint32_t left = je - (jl - 4u); // 4..7 tokens left, the first four already in the ring
            const uint32_t f = (uint32_t)fm;
            step(jl - 4u, f & 1u, ra);
            if (left > 4u)
                ra = row_at(o0, jl);
            step(jl - 3u, f & 2u, rb);
            if (left > 5u)
                rb = row_at(o1, jl);
            step(jl - 2u, f & 4u, rc);
            if (left > 6u)
                rc = row_at(o2, jl);
            step(jl - 1u, f & 8u, rd);
            if (left > 4u)
                step(jl, f & 16u, ra);
            if (left > 5u)
                step(jl + 1u, f & 32u, rb);
            if (left > 6u)
                step(jl + 2u, f & 64u, rc);
        }
    };

    WalkState A{ones, 0u, 0u, 0u}, B{ones, 0u, 0u, 0u};
    const uint32_t biasA = chainA * a.chain_rows, biasB = chainB * a.chain_rows;
    for (uint32_t tile = tile_begin; tile < tile_end; tile++, lane_ptr += a.in_tile_bytes)
    {
        A.acc = ones;
        B.acc = ones;
        A.sp = B.sp = 0u;
        run(std::false_type{}, tkA, cdA.ntok, 0u, cdA.ntok - k, biasA, A, stackA, A, stackA, preA, tokA);
        if (two)
            run(std::false_type{}, tkB, cdB.ntok, 0u, cdB.ntok - k, biasB, B, stackB, B, stackB, preB, tokB);
        if (k)
            run(std::true_type{}, tkA, cdA.ntok, cdA.ntok - k, cdA.ntok, biasA, A, stackA, B, stackB, true, tokA);
    }

    // the two lengths: changes of a lane = 32 sites per combine (and per chain start) minus the non-empty ones
    const uint32_t ntiles_here = tile_end - tile_begin;
    auto finish = [&](const CandDesc &cd, const WalkState &S, long long base, uint32_t cand) {
        uint32_t wsum = 32u * (cd.ncomb + cd.nfresh) * ntiles_here - S.nonempty - S.nonempty_rare;
        for (int off = 32; off > 0; off >>= 1)
            wsum += (uint32_t)__shfl_xor((int)wsum, off);
        unsigned long long total = wsum;
        if (group == 0)
            total += (unsigned long long)base;
        if (lane == 0)
            atomicAdd(a.len_out + cand, total + (WATCH ? 1ull << WATCH_COUNT_SHIFT : 0ull));
    };
    finish(cdA, A, baseA, ia);
    if (two)
        finish(cdB, B, baseB, ib);
}

// ---------------------------------------------------------------------------------------------
// small helpers around the walk

// changes[dst] = 0 for every node a commit program is about to recompute, plus the two scalars the
// commit walk accumulates into (the root slot of changes[] and the program's length slot): one launch
// instead of a launch and two memsets on the accept path
// ... and S_all gives up what those nodes had contributed (the walk adds the new counts back)
__global__ void zero_changes_kernel(unsigned long long *changes, const int32_t *dsts, uint32_t n,
                                    unsigned long long *root_slot, unsigned long long *len_slot,
                                    unsigned long long *s_all, uint32_t bias_from, uint32_t row_bias)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && dsts[i] >= 0)
    {
        const uint32_t slot = (uint32_t)dsts[i] >= bias_from ? (uint32_t)dsts[i] + row_bias : (uint32_t)dsts[i];
        const unsigned long long old = changes[slot];
        changes[slot] = 0ull;
        if (old)
            atomicAdd(s_all, 0ull - old); // two's complement: subtracts
    }
    if (i == 0)
    {
        *root_slot = 0ull;
        *len_slot = 0ull;
    }
}

// scalars[0] = sum of changes[first .. last) (all internal nodes of one resident tree); scalars[1] = that +
// changes[root_slot] (the root combines) = length of that tree.  One block.
__global__ __launch_bounds__(256) void sum_changes_kernel(const unsigned long long *changes, uint32_t first,
                                                          uint32_t last, uint32_t root_slot, long long *scalars)
{
    __shared__ long long part[4];
    long long s = 0;
    for (uint32_t i = first + threadIdx.x; i < last; i += 256u)
        s += (long long)changes[i];
    for (int off = 32; off > 0; off >>= 1)
        s += __shfl_xor(s, off);
    if ((threadIdx.x & 63u) == 0)
        part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        const long long all = part[0] + part[1] + part[2] + part[3];
        scalars[0] = all;
        scalars[1] = all + (long long)changes[root_slot];
    }
}

// fill the padding of every row (words [nwords, stride)) and whole rows [first_full_row, nrows)
// with all-ones: an all-N column never adds length and fitch(all, all) = all in either layout,
// so padded lanes stay inert
__global__ void fill_pad_kernel(uint64_t *rows, uint32_t nrows, uint32_t nwords, uint32_t stride_words,
                                uint32_t first_full_row)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= stride_words)
        return;
    for (uint32_t row = blockIdx.y; row < nrows; row += gridDim.y)
        if (w >= nwords || row >= first_full_row)
            rows[(size_t)row * stride_words + w] = ~0ull;
}

// in-place layout change of rows [0, nrows): to_planes ? reference nibbles -> bit planes : back
__global__ void relayout_kernel(uint4 *rows, uint32_t nrows, uint32_t stride4, uint32_t to_planes)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= stride4)
        return;
    for (uint32_t row = blockIdx.y; row < nrows; row += gridDim.y)
    {
        uint4 *p = rows + (size_t)row * stride4 + g;
        *p = to_planes ? nibbles_to_planes(*p) : planes_to_nibbles(*p);
    }
}

// one row of the tile-major resident block -> reference nibble layout, contiguous, in a scratch buffer (lvbgpu_get_sets)
__global__ void export_row_kernel(const uint4 *rows, uint32_t row, uint32_t total_rows, uint32_t ntiles, uint4 *out)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < ntiles * 64u)
        out[g] = planes_to_nibbles(rows[((size_t)(g >> 6) * total_rows + row) * 64u + (g & 63u)]);
}

// leaf rows as they are encoded (row-major) -> their places in the tile-major resident block
__global__ void rows_to_tiles_kernel(const uint4 *src, uint4 *dst, uint32_t nrows, uint32_t total_rows, uint32_t ntiles)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ntiles * 64u)
        return;
    for (uint32_t row = blockIdx.y; row < nrows; row += gridDim.y)
        dst[((size_t)(g >> 6) * total_rows + row) * 64u + (g & 63u)] = src[(size_t)row * ntiles * 64u + g];
}

// DNAToBinary on the device (reference DataOperations.c:164-249): one thread = one packed word in
// the reference's nibble layout.  text: n rows of m bytes (row-major, no terminators).  *bad
// becomes 1 + the smallest offending flat position if a symbol is not one the reference accepts.
__constant__ int8_t k_iupac[256];

__global__ void encode_text_kernel(const uint8_t *text, uint32_t n, uint64_t m, uint32_t nwords,
                                   uint32_t stride_words, uint64_t *rows, unsigned long long *bad)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nwords)
        return;
    for (uint32_t row = blockIdx.y; row < n; row += gridDim.y)
    {
        const uint8_t *src = text + (uint64_t)row * m;
        uint64_t w = 0;
        for (uint32_t k = 0; k < 16; k++)
        {
            const uint64_t site = (uint64_t)j * 16u + k;
            int s = 0xF; // padding is 'N' (DataOperations.c:187-188)
            if (site < m)
            {
                s = k_iupac[src[site]];
                if (s < 0)
                {
                    atomicMin(bad, (unsigned long long)row * m + site + 1ull);
                    s = 0xF;
                }
            }
            w |= (uint64_t)s << (4 * k);
        }
        rows[(size_t)row * stride_words + j] = w;
    }
}


// ---------------------------------------------------------------------------------------------
// Ceiling probe of the walk's memory path (lvbgpu_probe_l2): the walk's launch geometry and access pattern -
// one wave per (tile group, candidate), XCD-aware item order, `ntok` pseudo-random rows of the resident block
// per tile, RING row loads (1 KiB each) in flight - with the arithmetic replaced by one XOR per 16 bytes.
// What it reads per second is what the L2 -> CU path delivers to this pattern on this chip, measured in the
// run that quotes it.
template <int RING>
__global__ __launch_bounds__(WALK_THREADS) void l2_probe_kernel(const uint4 *rows, uint32_t stride4, uint32_t nrows,
                                                                uint32_t ntiles, uint32_t ngroups, uint32_t B,
                                                                uint32_t ntok, uint4 *sink)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nblk = gridDim.x;
    const uint32_t pos = (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3);
    const uint32_t item = pos * WALK_WAVES + wave;
    if (item >= B * ngroups)
        return;
    const uint32_t group = item / B, cand = item - group * B;
    const uint32_t t0 = (uint32_t)((uint64_t)group * ntiles / ngroups), t1 = (uint32_t)((uint64_t)(group + 1) * ntiles / ngroups);
    uint4 acc = make_uint4(0, 0, 0, 0);
    const uint32_t seed = cand * 2654435761u + 12345u;
    for (uint32_t t = t0; t < t1; t++)
    {
        // stride4 == 64: a TILE-MAJOR block ([tile][row][1 KiB], the resident layout); else row-major
        const uint4 *base = rows + (stride4 == 64u ? (size_t)t * nrows * 64u : (size_t)t * 64u) + lane;
        uint32_t s = seed;
        uint4 ring[RING];
#pragma unroll
        for (int q = 0; q < RING; q++)
        {
            s = s * 1664525u + 1013904223u;
            ring[q] = base[(size_t)((s >> 8) % nrows) * stride4];
        }
        for (uint32_t j = 0; j + RING <= ntok; j += RING)
        {
#pragma unroll
            for (int q = 0; q < RING; q++)
            {
                acc.x ^= ring[q].x;
                acc.y ^= ring[q].y;
                acc.z ^= ring[q].z;
                acc.w ^= ring[q].w;
                s = s * 1664525u + 1013904223u;
                ring[q] = base[(size_t)((s >> 8) % nrows) * stride4];
            }
        }
#pragma unroll
        for (int q = 0; q < RING; q++)
        {
            acc.x ^= ring[q].x;
            acc.y ^= ring[q].y;
            acc.z ^= ring[q].z;
            acc.w ^= ring[q].w;
        }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u && acc.z == 0x0fedcba9u) // never: keeps the loads alive
        sink[0] = acc;
}

hipError_t launch_l2_probe(const uint4 *rows, uint32_t stride4, uint32_t nrows, uint32_t ntiles, uint32_t ngroups,
                           uint32_t B, uint32_t ntok, int ring, uint4 *sink, uint64_t *loads_out, hipStream_t stream)
{
    if (B == 0 || ngroups == 0 || ngroups > ntiles || nrows == 0 || (uint64_t)B * ngroups >= (1ull << 31))
        return hipErrorInvalidValue;
    uint32_t nblk = (B * ngroups + WALK_WAVES - 1) / WALK_WAVES;
    nblk = (nblk + 7u) & ~7u;
    const dim3 grid(nblk), block(WALK_THREADS);
    if (ring == 8)
        hipLaunchKernelGGL(l2_probe_kernel<8>, grid, block, 0, stream, rows, stride4, nrows, ntiles, ngroups, B, ntok, sink);
    else
    {
        ring = 4;
        hipLaunchKernelGGL(l2_probe_kernel<4>, grid, block, 0, stream, rows, stride4, nrows, ntiles, ngroups, B, ntok, sink);
    }
    if (loads_out) // row loads (1 KiB each) one launch issues
        *loads_out = (uint64_t)B * ntiles * ((uint64_t)(ntok / (uint32_t)ring) * (uint32_t)ring + (uint32_t)ring);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// host-callable launchers (kernels.hpp)

static int8_t iupac_code(int c)
{
    // bit0=A bit1=C bit2=G bit3=T (LVB.h:73-76); table of DataOperations.c:190-232
    switch (c)
    {
    case 'A': return 1;
    case 'C': return 2;
    case 'G': return 4;
    case 'T': case 'U': return 8;
    case 'Y': return 2 | 8;
    case 'R': return 1 | 4;
    case 'W': return 1 | 8;
    case 'S': return 2 | 4;
    case 'K': return 8 | 4;
    case 'M': return 2 | 1;
    case 'B': return 2 | 4 | 8;
    case 'D': return 1 | 4 | 8;
    case 'H': return 1 | 2 | 8;
    case 'V': return 1 | 2 | 4;
    case 'N': case 'X': case '?': case '-': return 15;
    default: return -1;
    }
}

hipError_t upload_iupac_table()
{
    int8_t tab[256];
    for (int c = 0; c < 256; c++)
        tab[c] = iupac_code(c);
    return hipMemcpyToSymbol(HIP_SYMBOL(k_iupac), tab, sizeof(tab));
}

hipError_t shape_walk(WalkArgs &a, bool commit, uint32_t nwaves, size_t lds_budget, size_t *lds_out)
{
    const bool paired = a.pairs != nullptr && !commit;
    if (a.ngroups == 0 || a.ngroups > a.ntiles || a.nitems != (paired ? a.npairs : a.B) * a.ngroups || a.nitems >= (1u << 31) ||
        (paired && (a.npairs == 0 || (a.host_len && !a.watcher))))
        return hipErrorInvalidValue;
    a.tiles_per = a.ntiles / a.ngroups;
    a.tiles_rem = a.ntiles % a.ngroups;
    // floor(2^32 / B): mulhi(item, inv_B) is floor(item / B) or one less for item < 2^31 (the kernel
    // fixes up once).  B == 1 would need 2^32: 2^32 - 1 gives item - 1 (0 for item 0), same fix-up.
    {
        const uint32_t per_group = paired ? a.npairs : a.B; // items per tile group
        a.inv_B = per_group == 1u ? 0xFFFFFFFFu : (uint32_t)((1ull << 32) / per_group);
    }
    size_t lds = (size_t)nwaves * a.stack_depth * 64u * sizeof(uint4) * (paired ? 2u : 1u); // a paired wave keeps two operand stacks
    a.defer_slots = 0;
    if (commit)
    {
        // what LDS is left after the operand stack holds produced sets until a burst: 1 KiB per slot and wave
        // + 64 bytes for its lanes' change counts; at least one slot (check_depth leaves room for it)
        static const uint32_t max_slots = [] {
            const char *e = getenv("LVBGPU_DEFER_SLOTS"); // tests: small bursts
            const int v = e ? atoi(e) : 32;
            return (uint32_t)(v < 1 ? 1 : (v > 32 ? 32 : v));
        }();
        const uint32_t per_wave_kib = (uint32_t)(lds_budget / nwaves / 1024u);
        if (per_wave_kib < a.stack_depth + 2u)
            return hipErrorInvalidValue;
        const uint32_t room = (per_wave_kib - a.stack_depth) * 1024u / (1024u + 64u); // slots of 1 KiB + 64 B
        a.defer_slots = room < max_slots ? room : max_slots;
        lds += (size_t)nwaves * a.defer_slots * (64u * sizeof(uint4) + 64u);
    }
    if (lds > MAX_LDS_BYTES)
        return hipErrorInvalidValue;
    *lds_out = lds;
    return hipSuccess;
}

hipError_t launch_walk(const WalkArgs &args, bool commit, hipStream_t stream, uint32_t *flip_state)
{
    if (args.nitems == 0)
        return hipSuccess;
    const bool paired = args.pairs != nullptr && !commit;
    WalkArgs a = args;
    // alternate the direction of big scoring launches whose rows do not fit the L2s (LVBGPU_FLIP=0: never)
    static const bool allow_flip = [] {
        const char *e = getenv("LVBGPU_FLIP");
        return !(e && e[0] == '0');
    }();
    // (the direction alternates per CONTEXT - flip_state is the caller's counter: a process-wide one would be shared,
    // and raced on, by contexts driven from different host threads)
    a.flip = 0;
    if (allow_flip && flip_state && !commit && args.block_bytes > FLIP_MIN_BYTES)
        a.flip = ((*flip_state)++) & 1u;
    size_t lds = 0;
    {
        const hipError_t e = shape_walk(a, commit, WALK_WAVES, MAX_LDS_BYTES, &lds);
        if (e != hipSuccess)
            return e;
    }
    uint32_t nblk = (a.nitems + WALK_WAVES - 1) / WALK_WAVES;
    nblk = (nblk + 7u) & ~7u; // the XCD remap needs a multiple of 8
    // measurement knob: extra dynamic LDS per workgroup caps the waves a CU can hold (160 KiB per CU, 4 waves per
    // workgroup): LVBGPU_LDS_PAD_KB=36 -> 16 waves per CU, 76 -> 8
    static const size_t lds_pad = [] {
        const char *e = getenv("LVBGPU_LDS_PAD_KB");
        return e ? (size_t)atoi(e) * 1024u : (size_t)0;
    }();
    if (lds + lds_pad <= MAX_LDS_BYTES)
        lds += lds_pad;
    // offsets in 16-byte units must fit 32 bits; LVBGPU_WIDE_OFFSETS=1 forces the 64-bit form (tests)
    const bool wide = walk_needs_wide(args);
    if (commit || !a.host_len || a.ngroups > WATCH_MAX_GROUPS)
        a.watcher = 0;
    const dim3 grid(a.watcher ? nblk + 8u : nblk), block(WALK_THREADS);
    if (paired)
    {
        if (a.watcher)
        {
            if (wide)
                hipLaunchKernelGGL((fitch_walk_pair<true, true>), grid, block, lds, stream, a);
            else
                hipLaunchKernelGGL((fitch_walk_pair<false, true>), grid, block, lds, stream, a);
        }
        else if (wide)
            hipLaunchKernelGGL((fitch_walk_pair<true, false>), grid, block, lds, stream, a);
        else
            hipLaunchKernelGGL((fitch_walk_pair<false, false>), grid, block, lds, stream, a);
        return hipGetLastError();
    }
    if (commit)
    {
        if (wide)
            hipLaunchKernelGGL((fitch_walk<true, true>), grid, block, lds, stream, a);
        else
            hipLaunchKernelGGL((fitch_walk<true, false>), grid, block, lds, stream, a);
    }
    else if (a.watcher)
    {
        if (wide)
            hipLaunchKernelGGL((fitch_walk<false, true, 2>), grid, block, lds, stream, a);
        else
            hipLaunchKernelGGL((fitch_walk<false, false, 2>), grid, block, lds, stream, a);
    }
    else if (a.host_len)
    {
        if (wide)
            hipLaunchKernelGGL((fitch_walk<false, true, 1>), grid, block, lds, stream, a);
        else
            hipLaunchKernelGGL((fitch_walk<false, false, 1>), grid, block, lds, stream, a);
    }
    else if (wide)
        hipLaunchKernelGGL((fitch_walk<false, true>), grid, block, lds, stream, a);
    else
        hipLaunchKernelGGL((fitch_walk<false, false>), grid, block, lds, stream, a);
    return hipGetLastError();
}

bool walk_needs_wide(const WalkArgs &a)
{
    static const bool force_wide = [] {
        const char *e = getenv("LVBGPU_WIDE_OFFSETS");
        return e && e[0] == '1';
    }();
    return force_wide || (uint64_t)a.nrows * a.in_stride4 >= (1ull << 32);
}

// nothing but a clock watcher (lvbgpu_debug_stall: the wait-limit test needs a stream that is busy for a known time).
// One wave; wall_clock64 counts at 100 MHz; the loop is bounded twice over (clock and trip count).
__global__ __launch_bounds__(64) void stall_kernel(uint32_t ms)
{
    const uint64_t t0 = wall_clock64(), ticks = (uint64_t)ms * 100000ull;
    for (uint32_t trips = 0; trips < (1u << 26) && wall_clock64() - t0 < ticks; trips++)
        __builtin_amdgcn_s_sleep(64);
}

hipError_t launch_stall(hipStream_t stream, uint32_t ms)
{
    hipLaunchKernelGGL(stall_kernel, dim3(1), dim3(64), 0, stream, ms > 2000u ? 2000u : ms);
    return hipGetLastError();
}

hipError_t raise_lds_limit()
{
    // whole-tree programs may want more than the default 64 KiB of dynamic LDS
    for (const void *f : {reinterpret_cast<const void *>(&fitch_walk<true, true>),
                          reinterpret_cast<const void *>(&fitch_walk<true, false>),
                          reinterpret_cast<const void *>(&fitch_walk<false, true>),
                          reinterpret_cast<const void *>(&fitch_walk<false, false>),
                          reinterpret_cast<const void *>(&fitch_walk<false, true, 1>),
                          reinterpret_cast<const void *>(&fitch_walk<false, false, 1>),
                          reinterpret_cast<const void *>(&fitch_walk<false, true, 2>),
                          reinterpret_cast<const void *>(&fitch_walk<false, false, 2>),
                          reinterpret_cast<const void *>(&fitch_walk_pair<true, true>),
                          reinterpret_cast<const void *>(&fitch_walk_pair<false, true>),
                          reinterpret_cast<const void *>(&fitch_walk_pair<true, false>),
                          reinterpret_cast<const void *>(&fitch_walk_pair<false, false>)})
    {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS_BYTES);
        if (e != hipSuccess)
            return e;
    }
    return hipSuccess;
}

hipError_t launch_zero_changes(unsigned long long *changes, const int32_t *dsts, uint32_t n, unsigned long long *root_slot,
                               unsigned long long *len_slot, unsigned long long *s_all, uint32_t bias_from, uint32_t row_bias,
                               hipStream_t stream)
{
    hipLaunchKernelGGL(zero_changes_kernel, dim3(n ? (n + 255) / 256 : 1), dim3(256), 0, stream, changes, dsts, n, root_slot,
                       len_slot, s_all, bias_from, row_bias);
    return hipGetLastError();
}

hipError_t launch_sum_changes(const unsigned long long *changes, uint32_t first, uint32_t last, uint32_t root_slot,
                              long long *scalars, hipStream_t stream)
{
    hipLaunchKernelGGL(sum_changes_kernel, dim3(1), dim3(256), 0, stream, changes, first, last, root_slot, scalars);
    return hipGetLastError();
}

hipError_t launch_fill_pad(uint64_t *rows, uint32_t nrows, uint32_t nwords, uint32_t stride_words,
                           uint32_t first_full_row, hipStream_t stream)
{
    hipLaunchKernelGGL(fill_pad_kernel, dim3((stride_words + 255) / 256, nrows < 65535u ? nrows : 65535u), dim3(256), 0,
                       stream, rows, nrows, nwords, stride_words, first_full_row);
    return hipGetLastError();
}

hipError_t launch_relayout(uint4 *rows, uint32_t nrows, uint32_t stride4, bool to_planes, hipStream_t stream)
{
    if (nrows == 0)
        return hipSuccess;
    hipLaunchKernelGGL(relayout_kernel, dim3((stride4 + 255) / 256, nrows < 65535u ? nrows : 65535u), dim3(256), 0,
                       stream, rows, nrows, stride4, to_planes ? 1u : 0u);
    return hipGetLastError();
}

hipError_t launch_export_row(const uint4 *rows, uint32_t row, uint32_t total_rows, uint32_t ntiles, uint4 *out, hipStream_t stream)
{
    hipLaunchKernelGGL(export_row_kernel, dim3((ntiles * 64u + 255) / 256), dim3(256), 0, stream, rows, row, total_rows, ntiles, out);
    return hipGetLastError();
}

hipError_t launch_rows_to_tiles(const uint4 *src, uint4 *dst, uint32_t nrows, uint32_t total_rows, uint32_t ntiles, hipStream_t stream)
{
    if (nrows == 0)
        return hipSuccess;
    hipLaunchKernelGGL(rows_to_tiles_kernel, dim3((ntiles * 64u + 255) / 256, nrows < 65535u ? nrows : 65535u), dim3(256), 0, stream,
                       src, dst, nrows, total_rows, ntiles);
    return hipGetLastError();
}

hipError_t launch_encode_text(const uint8_t *text, uint32_t n, uint64_t m, uint32_t nwords, uint32_t stride_words,
                              uint64_t *rows, unsigned long long *bad, hipStream_t stream)
{
    hipLaunchKernelGGL(encode_text_kernel, dim3((nwords + 255) / 256, n < 65535u ? n : 65535u), dim3(256), 0, stream,
                       text, n, m, nwords, stride_words, rows, bad);
    return hipGetLastError();
}

} // namespace lvbgpu
