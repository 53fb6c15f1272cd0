// walk_body.hpp - the walk of ONE item (a candidate's postorder program over one tile group) by one wavefront: the
// body of fitch_walk (fitch_kernels.hip), shared with the commit role of the post launch (propose_kernels.hip), where a
// chain's accepted move is walked in commit form beside the table rebuild and the next step's generator.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

namespace lvbgpu
{

// ---------------------------------------------------------------------------------------------
// Fitch step on one 16-byte bit-plane group (32 sites).  x, y: child state sets (planes A,C,G,T
// in .x .y .z .w); returns the parent's set; adds the number of sites with a NON-empty
// intersection to `nonempty` (changes = 32 - that).  Per site this is the reference's word step
// (TreeEvaluation.c:219-230): intersection if non-empty, else union and one change.
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0xEA); // (a & b) | c
}
__device__ __forceinline__ uint32_t pick(uint32_t a, uint32_t b, uint32_t any)
{
    return __builtin_amdgcn_bitop3_b32(a, b, any, 0xD4); // any ? a & b : a | b
}

__device__ __forceinline__ uint4 fitch_planes(const uint4 x, const uint4 y, uint32_t &nonempty)
{
    uint32_t any = x.w & y.w;
    any = and_or(x.z, y.z, any);
    any = and_or(x.y, y.y, any);
    any = and_or(x.x, y.x, any);
    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(nonempty) : "v"(any)); // nonempty += popcount(any)
    uint4 z;
    z.x = pick(x.x, y.x, any);
    z.y = pick(x.y, y.y, any);
    z.z = pick(x.z, y.z, any);
    z.w = pick(x.w, y.w, any);
    return z;
}

// What the waves of a launch hand one another (partial lengths, partial change counts, arrival ticks) is written by
// agent-scope atomics and read by agent-scope atomic loads: those are performed where all XCDs see them, so the
// hand-over needs ORDER only - this wave's atomics acknowledged before its tick is sent - and no cache written back or
// invalidated.  __threadfence() does both (buffer_wbl2 of the XCD's whole L2 + buffer_inv): 3.5 us per wave on an idle
// chip, 6.5 with freshly written rows in the L2 (MI355X_MICROARCH.md), on every wave of a commit.  The same holds for
// what goes to the HOST before a flag (lengths, picked moves): written with system-scope atomic stores, which are
// written through - once acknowledged they are in no cache of this chip, and the flag is sent after that.
__device__ __forceinline__ void atomics_acknowledged()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------
// The walk.  COMMIT: store every produced node set to rows_out[dst] and add its change count to
// changes_out[dst] (accepting a candidate / full evaluation / strict-compat write-back).
typedef const __attribute__((address_space(1))) char *global_cp; // keeps loads global_load, not flat_load
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct OffVec
{
    uint32_t lo, hi; // one 64-bit value per lane, split over two VGPRs
};

// One item = (tile group, candidate).  lds_stack: the workgroup's dynamic LDS ([wave][level][lane] operand stacks, then
// the commit form's parked sets and counts); wave / nwaves: this wave's place in its workgroup; item < a.nitems.
// COMMIT: store every produced node set to rows_out[dst] and add its change count to changes_out[dst] (accepting a
// candidate / full evaluation / strict-compat write-back).  HANDOVER: how a scoring launch's lengths reach the host - 0:
// they stay in len_out (the caller copies them); 1: direct step (small launches: the last wave, or with one wave per
// candidate every wave, stores them to the host); 2: watcher waves (the walking waves count themselves into the slot).
template <bool COMMIT, bool WIDE, int HANDOVER>
__device__ __forceinline__ void walk_item(const WalkArgs &a, uint4 *const lds_stack, const uint32_t lane, const uint32_t wave,
                                          const uint32_t nwaves, const uint32_t item)
{
    constexpr bool WATCH = HANDOVER == 2, DIRECT = HANDOVER == 1;
    static_assert(!(COMMIT && HANDOVER != 0), "scoring launches only");
    // an item = (tile group, candidate): the wave walks the candidate's program once per tile of
    // its group, so descriptor/token fetches and the final reduction are paid once per group
    // (no division: a wave's fixed cost is scalar work too)
    auto split = [&](uint32_t it, uint32_t &grp, uint32_t &cnd) {
        grp = __umulhi(it, a.inv_B); // floor(item / B) or one less
        cnd = it - grp * a.B;
        if (cnd >= a.B)
        {
            grp++;
            cnd -= a.B;
        }
    };
    uint32_t group, cand;
    split(item, group, cand);
    const uint32_t tile_begin = group * a.tiles_per + (group < a.tiles_rem ? group : a.tiles_rem);
    const uint32_t tile_end = tile_begin + a.tiles_per + (group < a.tiles_rem ? 1u : 0u);

    // COMMIT: launch candidate `cand` is cands[pick_idx[cand]] (use_pick: the accepted candidates of a device-built batch),
    // and from n_first on candidate cand - n_first of a SECOND program block (cands2 / toks2 / dsts2: host-built programs,
    // e.g. re-roots, committed in the same launch)
    uint32_t which = cand;
    const CandDesc *cands = a.cands;
    const uint32_t *toks = a.toks;
    const int32_t *dsts = a.dsts;
    if constexpr (COMMIT)
    {
        if (cand >= a.n_first)
        {
            which = cand - a.n_first;
            cands = a.cands2;
            toks = a.toks2;
            dsts = a.dsts2;
        }
        else if (a.use_pick)
            which = a.pick_idx[cand];
    }
    const CandDesc cd = cands[which];
    // which resident tree: node numbers from bias_from on (the internal nodes) move by the chain's row block
    const uint32_t chain = cd.flags >> CAND_CHAIN_SHIFT;
    const uint32_t row_bias = chain * a.chain_rows;
    auto biased = [&](uint32_t v) { return v >= a.bias_from ? v + row_bias : v; };
    const uint32_t *__restrict__ tk = toks + cd.tok_off;
    const int32_t *__restrict__ ds = dsts + cd.dst_off;
    // this lane's 16-byte group of row 0 in the wave's first tile: input as a byte address (lane_ptr below), output as
    // an index in 16-byte units (kernels.hpp: tile-major resident block, row-major staging arenas)
    [[maybe_unused]] uint64_t out_col = (uint64_t)tile_begin * a.out_tile4 + lane;
    const char *__restrict__ in = reinterpret_cast<const char *>(a.rows_in) + (uint64_t)tile_begin * a.in_tile_bytes + lane * 16u;

    uint4 acc;
    uint32_t sp = 0;       // stack pointer (levels)
    uint32_t nonempty = 0; // sites with non-empty intersection, this lane, whole program
    [[maybe_unused]] uint32_t k_comb = 0; // COMMIT: combines done in this tile (index into ds[])
    uint4 *const my_stack = lds_stack + (size_t)wave * a.stack_depth * 64u + lane;

    // row at byte offset `off` (wave-uniform) from rows_in -> this lane's group.  lane_ptr (rows_in +
    // this lane's column offset) is kept opaque so the add stays ONE vector instruction taking the
    // scalar pair as an operand, instead of being regrouped into scalar adds
    global_cp lane_ptr = (global_cp)in;
    auto load_row = [&](uint64_t off) -> uint4 {
        asm volatile("" : "+v"(lane_ptr));
        const u32x4 v = *reinterpret_cast<const __attribute__((address_space(1))) u32x4 *>(lane_ptr + off);
        return make_uint4(v.x, v.y, v.z, v.w);
    };
    // narrow form: off16 counts 16-byte units; v_lshl_add_u64 shifts it on the way into the add
    auto load_row16 = [&](uint32_t off16) -> uint4 {
        asm volatile("" : "+v"(lane_ptr));
        const u32x4 v = *reinterpret_cast<const __attribute__((address_space(1))) u32x4 *>(lane_ptr + ((uint64_t)off16 << 4));
        return make_uint4(v.x, v.y, v.z, v.w);
    };

    // COMMIT with defer_slots: [slot][lane] sets and [slot][lane] change counts (one byte each: <= 32 sites per lane) of
    // the combines not yet written out.  The counts are summed per combine at the burst, 64 bytes per lane, instead of
    // with six ballots per token: a commit is a handful of lone waves whose time is their instruction latency
    // (25 of a token's ~65 instructions were that sum).
    uint4 *const my_rows = lds_stack + (size_t)nwaves * a.stack_depth * 64u + (size_t)wave * a.defer_slots * 64u + lane;
    // (an LDS-typed pointer: through a generic uint8_t * the byte stores may alias anything - the compiler then keeps
    // the walk's state in scratch around each of them)
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    lds_u8 *const my_cnt = (lds_u8 *)(reinterpret_cast<uint8_t *>(lds_stack + (size_t)nwaves * (a.stack_depth + a.defer_slots) * 64u) +
                                      (size_t)wave * a.defer_slots * 64u);
    uint32_t pend = 0;      // combines waiting in LDS
    uint32_t k_flushed = 0; // combines of this tile already written out
    auto add_count = [&](uint32_t k, int32_t dst, uint32_t s) __attribute__((always_inline)) {
        if (a.tmp_changes)
            atomicAdd(a.tmp_changes + (size_t)cand * a.tmp_stride + k, (unsigned long long)s); // settled by the candidate's last wave
        else
        {
            atomicAdd(a.changes_out + (dst >= 0 ? biased((uint32_t)dst) : a.root_slot + chain), (unsigned long long)s);
            if (dst >= 0 && a.s_all_out)
                atomicAdd(a.s_all_out + 4u * chain, (unsigned long long)s); // S_all follows the commit: no separate summing pass
        }
    };
    // write out what waits in LDS: destinations fetched with one vector load BEFORE the first store, so the
    // burst itself never waits on memory; lane s adds combine s's count (at most 64 slots)
    auto flush = [&]() __attribute__((always_inline)) {
        if (pend == 0)
            return;
        const int32_t mydst = lane < pend ? ds[k_flushed + lane] : -1;
        __builtin_amdgcn_wave_barrier(); // (compiler only: the lanes' byte stores before other lanes' reads of them)
        uint32_t mycnt = 0;
        if (lane < pend)
        {
            const lds_u32 *const w = (const lds_u32 *)(my_cnt + (size_t)lane * 64u);
#pragma unroll
            for (int q = 0; q < 16; q++)
                mycnt = __builtin_amdgcn_udot4(w[q], 0x01010101u, mycnt, false);
        }
        for (uint32_t s = 0; s < pend; s++)
        {
            const int32_t dst = __builtin_amdgcn_readlane(mydst, (int)s);
            if (dst >= 0)
                a.rows_out[(size_t)biased((uint32_t)dst) * a.out_stride4 + out_col] = my_rows[(size_t)s * 64u];
        }
        if (mycnt)
            add_count(k_flushed + lane, mydst, mycnt);
        k_flushed += pend;
        pend = 0;
        // drain here, once per burst: with a write possibly in flight the compiler would turn every later wait
        // for a row into vmcnt(0) (reads and writes complete out of order with respect to each other)
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0), expcnt and lgkmcnt untouched
    };
    // COMMIT: the combine just done produced node ds[k_comb] with `ch` changes in this lane; set and count wait
    // in LDS for the next burst
    auto produce = [&](uint32_t ch) __attribute__((always_inline)) {
        my_rows[(size_t)pend * 64u] = acc;
        my_cnt[(size_t)pend * 64u + lane] = (uint8_t)ch; // ch <= 32
        pend++;
        k_comb++;
        if (__builtin_expect(pend == a.defer_slots, 0))
            flush();
    };

    // One token = one combine, acc = fitch(acc, row), and ONE scalar test: 15 instructions, of which the load
    // is what the loop waits for.  Scalar instructions are the ones that cost when added (tools/l2_probe3.hip:
    // beyond ~4 per 1 KiB load the read rate falls, VALU work up to 16 per load is free), so everything rare is
    // folded behind a single precomputed bit per token:
    //   * a chain start (FRESH) needs acc = row.  Instead of testing for it, the token BEFORE it
    //     leaves acc = all-ones (after pushing the old acc if the chain start says PUSH): then the
    //     ordinary combine gives fitch(all, row) = row and counts 32 non-empty sites, which cd.nfresh
    //     accounts for.  acc starts all-ones for the program's first token.
    //   * merges after a token's own step are rare as well.
    //   post bit j = token j has merges, or token j+1 is a chain start.
    const uint4 ones = make_uint4(~0u, ~0u, ~0u, ~0u);
    uint32_t nonempty_rare = 0; // kept apart so the hot path's counter has a single definition

    // what the length's base needs (the cached changes of this candidate's dirty nodes, S_all) is requested
    // here, together with the first rows, instead of as three dependent round trips after the walk - a small
    // launch is a chain of round trips and nothing else
    long long sub_early = 0, s_all_early = 0;
    if (group == 0 && (cd.flags & CAND_RESIDENT_BASE))
    {
        for (uint32_t i = lane; i < cd.ncomb; i += 64u)
        {
            const int32_t dst = ds[i];
            if (dst >= 0)
                sub_early += a.node_changes[biased((uint32_t)dst)];
        }
        s_all_early = a.s_all[4u * chain];
    }

    for (uint32_t tile = tile_begin; tile < tile_end; tile++, out_col += a.out_tile4, lane_ptr += a.in_tile_bytes)
    {
        acc = ones;
        if constexpr (COMMIT)
        {
            k_comb = 0;
            k_flushed = 0;
        }
        for (uint32_t c0 = 0; c0 < cd.ntok; c0 += 64u)
        {
            const uint32_t cnt = (cd.ntok - c0 < 64u) ? cd.ntok - c0 : 64u;
            // lane k holds token c0+k and that row's offset: one coalesced load + one multiply for 64 tokens
            const uint32_t mytok = (lane < cnt) ? tk[c0 + lane] : 0u;
            // that row's offset: bytes in two vectors (WIDE) or 16-byte units in one
            const uint32_t myrow = biased(mytok & TOK_ROW_MASK);
            const uint64_t myoff64 = WIDE ? (uint64_t)myrow * ((uint64_t)a.in_stride4 << 4)
                                          : (uint64_t)(myrow * a.in_stride4); // bytes | 16-byte units
            OffVec o0{(uint32_t)myoff64, (uint32_t)(myoff64 >> 32)};
            // the same offsets seen from 1, 2, 3 lanes further down, so that the four refills of a
            // group read lane j of four vectors with ONE scalar index
            auto down = [&](const OffVec &v, uint32_t k) {
                const int sel = (int)(((lane + k) & 63u) << 2);
                return OffVec{(uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)v.lo),
                              WIDE ? (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)v.hi) : 0u};
            };
            const OffVec o1 = down(o0, 1u), o2 = down(o0, 2u), o3 = down(o0, 3u);
            auto row_at = [&](const OffVec &v, uint32_t j) -> uint4 {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)v.lo, (int)j);
                if constexpr (!WIDE)
                    return load_row16(lo);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)v.hi, (int)j);
                return load_row(((uint64_t)hi << 32) | lo);
            };
            const uint64_t freshm = __builtin_amdgcn_ballot_w64((mytok & TOK_FRESH) != 0u);
            const uint64_t mergem = __builtin_amdgcn_ballot_w64(((mytok >> TOK_MERGE_SHIFT) & TOK_MERGE_MASK) != 0u);
            const uint64_t postm = mergem | (freshm >> 1);
            auto tok_at = [&](uint32_t j) { return (uint32_t)__builtin_amdgcn_readlane((int)mytok, (int)j); };

            // rare: what follows token j's own combine
            auto post = [&](uint32_t j) __attribute__((always_inline)) {
                const uint32_t tok = tok_at(j);
                for (uint32_t m = (tok >> TOK_MERGE_SHIFT) & TOK_MERGE_MASK; m != 0; m--)
                {
                    sp--;
                    const uint4 other = my_stack[(size_t)sp * 64u];
                    const uint32_t b2 = nonempty_rare;
                    acc = fitch_planes(other, acc, nonempty_rare);
                    if constexpr (COMMIT)
                        produce(32u - (nonempty_rare - b2));
                }
                if (j < 63u && ((freshm >> (j + 1u)) & 1u)) // the next token (of this chunk) starts a chain
                {
                    if (tok_at(j + 1u) & TOK_PUSH)
                    {
                        my_stack[(size_t)sp * 64u] = acc;
                        sp++;
                    }
                    acc = ones;
                }
            };
            auto step = [&](uint32_t j, uint32_t flagged, const uint4 cur) __attribute__((always_inline)) {
                const uint32_t before = nonempty;
                acc = fitch_planes(acc, cur, nonempty);
                if constexpr (COMMIT)
                {
                    if (!((freshm >> j) & 1u))
                        produce(32u - (nonempty - before));
                }
                if (__builtin_expect(flagged != 0u, 0))
                    post(j);
            };
            if (c0 != 0u && (freshm & 1u)) // a chunk of a long program that opens with a chain start
            {
                if (tok_at(0) & TOK_PUSH)
                {
                    my_stack[(size_t)sp * 64u] = acc;
                    sp++;
                }
                acc = ones;
            }

            if (cnt < 4u)
            {
                for (uint32_t j = 0; j < cnt; j++)
                    step(j, (uint32_t)(postm >> j) & 1u, row_at(o0, j));
                continue;
            }
            // 4-slot ring: slot q holds the row of token j+q
            uint4 ra = row_at(o0, 0), rb = row_at(o0, 1), rc = row_at(o0, 2), rd = row_at(o0, 3);
            uint32_t jl = 4;     // first token not yet in the ring
            uint64_t fm = postm; // post bits of the tokens from the head of the ring on, bit 0 first
#define LVB_GROUP(F, J0, JL)                                                                                  \
    step((J0), (F) & 1u, ra);                                                                                 \
    ra = row_at(o0, (JL));                                                                                    \
    step((J0) + 1u, (F) & 2u, rb);                                                                            \
    rb = row_at(o1, (JL));                                                                                    \
    step((J0) + 2u, (F) & 4u, rc);                                                                            \
    rc = row_at(o2, (JL));                                                                                    \
    step((J0) + 3u, (F) & 8u, rd);                                                                            \
    rd = row_at(o3, (JL));
            // every refill below is in range: no branches, counted vmcnt.  Two groups per trip to
            // halve the loop's own scalar instructions.
            for (; jl + 8u <= cnt; jl += 8u, fm >>= 8)
            {
                const uint32_t f = (uint32_t)fm;
                LVB_GROUP(f, jl - 4u, jl)
                LVB_GROUP(f >> 4, jl, jl + 4u)
            }
            if (jl + 4u <= cnt)
            {
                const uint32_t f = (uint32_t)fm;
                LVB_GROUP(f, jl - 4u, jl)
                jl += 4u;
                fm >>= 4;
            }
#undef LVB_GROUP
            // 4..7 tokens left, the first four already in the ring
            const uint32_t left = cnt - (jl - 4u);
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
        if constexpr (COMMIT)
            flush(); // this tile's last sets (out_col moves on with the tile)
    }

    // changes of this lane = 32 sites per combine (and per chain start, see step) minus the non-empty ones
    const uint32_t per_lane = 32u * (cd.ncomb + cd.nfresh) * (tile_end - tile_begin);
    const uint32_t lane_changes = per_lane - nonempty - nonempty_rare;
    // butterfly over the LDS crossbar: no scalar instructions (the bit-sliced ballot sum that
    // produce() uses for its 6-bit counts would cost ~4 per bit here)
    uint32_t wsum = lane_changes;
    for (int off = 32; off > 0; off >>= 1)
        wsum += (uint32_t)__shfl_xor((int)wsum, off);
    unsigned long long total = wsum;

    if (group == 0)
    {
        // clean nodes contribute their cached changes (TreeEvaluation.c:191-202):
        // base = cd.base + [resident: S_all - sum over this candidate's dirty nodes of changes]
        long long base = cd.base;
        if (cd.flags & CAND_RESIDENT_BASE)
        {
            long long sub = sub_early;
            for (int off = 32; off > 0; off >>= 1)
                sub += __shfl_xor(sub, off);
            base += s_all_early - sub;
        }
        total += (unsigned long long)base;
    }
    if constexpr (!COMMIT)
    {
        if (DIRECT && a.ngroups == 1u)
        {
            // direct step, one wave per candidate: `total` is the whole length - straight to the host, and the
            // last wave to tick releases the flag (every wave's store is system-visible before its tick)
            if (lane == 0)
            {
                __hip_atomic_store(a.host_len + cand, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                atomics_acknowledged(); // (a system-scope store is written through: acknowledged = out of every cache)
                if (atomicAdd(a.done_count, 1u) == a.nitems - 1u)
                {
                    __hip_atomic_store(a.done_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    atomics_acknowledged();
                    __hip_atomic_store(a.host_flag, a.step_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            return;
        }
    }
    if (lane == 0 && !(COMMIT && a.tmp_changes)) // a fused commit's length is S_all + the root slot
        atomicAdd(a.len_out + cand, total + (WATCH ? 1ull << WATCH_COUNT_SHIFT : 0ull));

    if constexpr (COMMIT)
    {
        if (a.tmp_changes)
        {
            // every candidate of the launch is walked by ngroups waves: the last of them to finish settles it
            uint32_t last = 0;
            unsigned long long *const tmp = a.tmp_changes + (size_t)cand * a.tmp_stride;
            if (lane == 0)
            {
                atomics_acknowledged(); // our partial counts before our tick
                last = atomicAdd(a.done_count + cand, 1u) == a.ngroups - 1u ? 1u : 0u;
            }
            if (__builtin_amdgcn_readfirstlane(last))
            {
                // (the tick has returned: every other wave's counts were acknowledged before its own tick was sent)
                long long delta = 0;               // new - old over the recomputed internal nodes
                unsigned long long root_changes = 0; // the two root combines (dst < 0)
                for (uint32_t i = lane; i < cd.ncomb; i += 64u)
                {
                    const unsigned long long v = __hip_atomic_load(tmp + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(tmp + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int32_t dst = ds[i];
                    if (dst >= 0)
                    {
                        const uint32_t slot = biased((uint32_t)dst);
                        delta += (long long)v - (long long)a.changes_out[slot];
                        a.changes_out[slot] = v;
                    }
                    else
                        root_changes += v;
                }
                for (int off = 32; off > 0; off >>= 1)
                {
                    delta += __shfl_xor(delta, off);
                    root_changes += __shfl_xor(root_changes, off);
                }
                if (lane == 0)
                {
                    a.changes_out[a.root_slot + chain] = root_changes;
                    a.s_all_out[4u * chain] += (unsigned long long)delta;
                    __hip_atomic_store(a.done_count + cand, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
    else
    {
        if constexpr (DIRECT)
        {
            // the launch's last wave hands the lengths to the host itself (threadfence-reduction pattern)
            uint32_t last = 0;
            if (lane == 0)
            {
                atomics_acknowledged(); // our sum before our count
                last = atomicAdd(a.done_count, 1u) == a.nitems - 1u ? 1u : 0u;
            }
            if (__builtin_amdgcn_readfirstlane(last))
            {
                // (every other wave's sum was acknowledged before its count was sent, and ours has returned)
                for (uint32_t i = lane; i < a.B; i += 64u)
                {
                    const unsigned long long v = __hip_atomic_load(a.len_out + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(a.len_out + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next step
                    __hip_atomic_store(a.host_len + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                if (lane == 0)
                    __hip_atomic_store(a.done_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomics_acknowledged(); // the wave's stores (all lanes: one counter per wave) before the flag
                if (lane == 0)
                    __hip_atomic_store(a.host_flag, a.step_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

} // namespace lvbgpu
