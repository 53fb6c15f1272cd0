// fitch_kernels.hip - hand-written gfx950 (MI355X, CDNA4) kernels of the Fitch scoring path.
//
// What is computed is exactly getplen's arithmetic (reference TreeEvaluation.c:204-264): for a
// list of dirty nodes in postorder, z = fitch(x, y) per packed word with the number of sites
// that needed a union counted, plus the two root combines.  How it is laid out is native:
//
//   * one wavefront (64 lanes) walks one candidate's postorder program for one 128-word site
//     tile: a lane owns 16 bytes (two 64-bit reference words = 32 sites) so every state-set row
//     load is one coalesced 1 KiB global_load_dwordx4 per wave;
//   * the running node set lives in 4 VGPRs ("acc"); sibling sets that must wait for the other
//     subtree sit on an operand stack - two register levels for NNI/SPR/TBR deltas, LDS
//     (ds_write_b128 / ds_read_b128, one 1 KiB level per wave) for whole-tree programs;
//   * the 64-bit SWAR step is split into independent 32-bit halves (no carry crosses a nibble,
//     so the split is exact): 11 VALU ops per half incl. v_bcnt_u32_b32 accumulation and one
//     v_bfi_b32 select;
//   * union counts are reduced across the wave with ballot + scalar popcount bit-slices (all
//     SALU, result lands in an SGPR) and added with one integer atomic per (candidate, tile):
//     integer addition commutes, so lengths are bit-exact whatever the arrival order;
//   * blockIdx -> work mapping is XCD-aware: items are ordered tile-major and each of the 8 XCDs
//     takes a contiguous eighth, so the waves resident on one XCD at any moment read the same
//     column slice of the resident tree and hit that XCD's private 4 MiB L2.
//
// No MFMA: this is bitwise integer streaming, bound by L2/HBM bandwidth and VALU issue.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

namespace lvbgpu
{

// ---------------------------------------------------------------------------------------------
// Fitch step on one 32-bit half (8 sites).  x, y: child state sets; returns the parent's set and
// adds the number of sites with a NON-empty intersection to `nonempty` (changes = 8 - that).
// Reference: TreeEvaluation.c:219-230 with MASK_SEVEN/MASK_EIGHT of LVB.h:88-89.
__device__ __forceinline__ uint32_t fitch32(uint32_t x, uint32_t y, uint32_t &nonempty)
{
    const uint32_t M7 = 0x77777777u, M8 = 0x88888888u;
    const uint32_t both = x & y;
    const uint32_t either = x | y;
    uint32_t u = (both & M7) + M7;      // bit 3 of a nibble: its low three bits are not all zero
    u = (u | both) & M8;                // ... or bit 3 itself is set  => intersection non-empty
    nonempty += __builtin_popcount(u);  // v_bcnt_u32_b32 accumulates
    const uint32_t full = (u << 1) - (u >> 3); // 0xF in every non-empty nibble (mod 2^32 is exact)
    // non-empty: keep the intersection; empty: take the union          (one v_bfi_b32)
    return (full & both) | (~full & either);
}

__device__ __forceinline__ uint4 fitch128(const uint4 x, const uint4 y, uint32_t &nonempty)
{
    uint4 z;
    z.x = fitch32(x.x, y.x, nonempty);
    z.y = fitch32(x.y, y.y, nonempty);
    z.z = fitch32(x.z, y.z, nonempty);
    z.w = fitch32(x.w, y.w, nonempty);
    return z;
}

// Sum of a small per-lane value over the 64 lanes of the wave, computed on the scalar unit:
// bit b of the sum's binary expansion is weighted popcount(ballot(bit b of v)).
__device__ __forceinline__ uint32_t wave_sum_bits(uint32_t v, uint32_t nbits)
{
    uint32_t total = 0;
    for (uint32_t b = 0; b < nbits; b++)
    {
        const uint64_t m = __builtin_amdgcn_ballot_w64(((v >> b) & 1u) != 0u);
        total += (uint32_t)__builtin_popcountll(m) << b;
    }
    return total;
}

// ---------------------------------------------------------------------------------------------
// The walk.  LDS_STACK: operand stack in LDS (any depth) instead of two register levels.
// COMMIT: store every produced node set to rows_out[dst] and add its change count to
// changes_out[dst] (accepting a candidate / full evaluation / strict-compat write-back).
template <bool LDS_STACK, bool COMMIT>
__global__ __launch_bounds__(WALK_THREADS) void fitch_walk(const WalkArgs a)
{
    extern __shared__ uint4 lds_stack[]; // [wave][level][lane], LDS_STACK only

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // XCD-aware remap: hardware deals consecutive block ids round-robin over the 8 XCDs; give
    // XCD x the x-th contiguous eighth of the tile-major item list.  gridDim.x % 8 == 0.
    const uint32_t nblk = gridDim.x;
    const uint32_t pos = (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3);
    const uint32_t item = pos * WALK_WAVES + wave;
    if (item >= a.nitems)
        return;
    const uint32_t tile = item / a.B;
    const uint32_t cand = item - tile * a.B;

    const CandDesc cd = a.cands[cand];
    const uint32_t *__restrict__ tk = a.toks + cd.tok_off;
    const int32_t *__restrict__ ds = a.dsts + cd.dst_off;
    const size_t col = (size_t)tile * 64u + lane;
    const uint4 *__restrict__ in = a.rows_in + col;

    uint4 acc = make_uint4(0, 0, 0, 0), s0 = acc, s1 = acc;
    uint32_t sp = 0;        // LDS stack pointer (levels)
    uint32_t nonempty = 0;  // sites with non-empty intersection, this lane, whole program
    uint32_t k_comb = 0;    // combines done (index into ds[])
    uint4 *const my_stack = lds_stack + (size_t)wave * a.stack_depth * 64u + lane;

    auto produce = [&](uint32_t ne_before) {
        if constexpr (COMMIT)
        {
            const int32_t dst = ds[k_comb];
            const uint32_t ch = 32u - (nonempty - ne_before); // <= 32: 6 bits
            const uint32_t s = wave_sum_bits(ch, 6);
            if (dst >= 0)
                a.rows_out[(size_t)dst * a.out_stride4 + col] = acc;
            if (lane == 0 && s)
                atomicAdd(a.changes_out + (dst >= 0 ? (uint32_t)dst : a.root_slot), (unsigned long long)s);
        }
        k_comb++;
    };

    uint32_t tok = tk[0];
    uint4 cur = in[(size_t)(tok & TOK_ROW_MASK) * a.in_stride4];
    for (uint32_t k = 0; k < cd.ntok; k++)
    {
        // prefetch the next token's row before working on this one: the row stream depends only
        // on the program, never on computed values
        const uint32_t tok_next = tk[(k + 1 < cd.ntok) ? k + 1 : k];
        const uint4 nxt = in[(size_t)(tok_next & TOK_ROW_MASK) * a.in_stride4];

        if (tok & TOK_FRESH)
        {
            if (tok & TOK_PUSH)
            {
                if constexpr (LDS_STACK)
                {
                    my_stack[(size_t)sp * 64u] = acc;
                    sp++;
                }
                else
                {
                    s1 = s0;
                    s0 = acc;
                }
            }
            acc = cur;
        }
        else
        {
            const uint32_t before = nonempty;
            acc = fitch128(acc, cur, nonempty);
            produce(before);
        }
        for (uint32_t m = (tok >> TOK_MERGE_SHIFT) & TOK_MERGE_MASK; m != 0; m--)
        {
            uint4 other;
            if constexpr (LDS_STACK)
            {
                sp--;
                other = my_stack[(size_t)sp * 64u];
            }
            else
            {
                other = s0;
                s0 = s1;
            }
            const uint32_t before = nonempty;
            acc = fitch128(other, acc, nonempty);
            produce(before);
        }
        tok = tok_next;
        cur = nxt;
    }

    // changes of this lane = 32 sites per combine minus the non-empty ones
    const uint32_t lane_changes = 32u * cd.ncomb - nonempty;
    const uint32_t nbits = 32u - __builtin_clz(32u * cd.ncomb | 1u);
    unsigned long long total = wave_sum_bits(lane_changes, nbits);

    if (tile == 0)
    {
        // clean nodes contribute their cached changes (TreeEvaluation.c:191-202):
        // base = cd.base + [resident: S_all - sum over this candidate's dirty nodes of changes]
        long long base = cd.base;
        if (cd.flags & CAND_RESIDENT_BASE)
        {
            long long sub = 0;
            for (uint32_t i = lane; i < cd.ncomb; i += 64u)
            {
                const int32_t dst = ds[i];
                if (dst >= 0)
                    sub += a.node_changes[dst];
            }
            for (int off = 32; off > 0; off >>= 1)
                sub += __shfl_xor(sub, off);
            base += *a.s_all - sub;
        }
        total += (unsigned long long)base;
    }
    if (lane == 0)
        atomicAdd(a.len_out + cand, total);
}

// ---------------------------------------------------------------------------------------------
// small helpers around the walk

// changes[dst] = 0 for every node a commit program is about to recompute
__global__ void zero_changes_kernel(unsigned long long *changes, const int32_t *dsts, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && dsts[i] >= 0)
        changes[dsts[i]] = 0ull;
}

// scalars[0] = sum of changes[first .. last) (all internal nodes); scalars[1] = that + changes[last]
// (the root combines) = length of the resident tree.  One block.
__global__ __launch_bounds__(256) void sum_changes_kernel(const unsigned long long *changes, uint32_t first,
                                                          uint32_t last, long long *scalars)
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
        scalars[1] = all + (long long)changes[last];
    }
}

// fill the padding of every row (words [nwords, stride)) and whole rows [first_row, nrows) with
// all-ones: an all-N column never adds length and fitch(F, F) = F, so padded lanes stay inert
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

// DNAToBinary on the device (reference DataOperations.c:164-249): one thread = one packed word.
// text: n rows of m bytes (row-major, no terminators).  *bad is set to 1 + the first offending
// flat position seen by some thread if a symbol is not one the reference accepts.
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

hipError_t launch_walk(const WalkArgs &a, bool lds_stack, bool commit, hipStream_t stream)
{
    if (a.nitems == 0)
        return hipSuccess;
    uint32_t nblk = (a.nitems + WALK_WAVES - 1) / WALK_WAVES;
    nblk = (nblk + 7u) & ~7u; // the XCD remap needs a multiple of 8
    const size_t lds = lds_stack ? (size_t)WALK_WAVES * a.stack_depth * 64u * sizeof(uint4) : 0;
    const dim3 grid(nblk), block(WALK_THREADS);
    if (lds_stack)
    {
        if (commit)
            hipLaunchKernelGGL((fitch_walk<true, true>), grid, block, lds, stream, a);
        else
            hipLaunchKernelGGL((fitch_walk<true, false>), grid, block, lds, stream, a);
    }
    else
    {
        if (commit)
            hipLaunchKernelGGL((fitch_walk<false, true>), grid, block, 0, stream, a);
        else
            hipLaunchKernelGGL((fitch_walk<false, false>), grid, block, 0, stream, a);
    }
    return hipGetLastError();
}

hipError_t raise_lds_limit()
{
    // whole-tree programs may want more than the default 64 KiB of dynamic LDS
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fitch_walk<true, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS_BYTES);
    if (e != hipSuccess)
        return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&fitch_walk<true, false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS_BYTES);
}

hipError_t launch_zero_changes(unsigned long long *changes, const int32_t *dsts, uint32_t n, hipStream_t stream)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(zero_changes_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, changes, dsts, n);
    return hipGetLastError();
}

hipError_t launch_sum_changes(const unsigned long long *changes, uint32_t first, uint32_t last, long long *scalars,
                              hipStream_t stream)
{
    hipLaunchKernelGGL(sum_changes_kernel, dim3(1), dim3(256), 0, stream, changes, first, last, scalars);
    return hipGetLastError();
}

hipError_t launch_fill_pad(uint64_t *rows, uint32_t nrows, uint32_t nwords, uint32_t stride_words,
                           uint32_t first_full_row, hipStream_t stream)
{
    hipLaunchKernelGGL(fill_pad_kernel, dim3((stride_words + 255) / 256, nrows < 65535u ? nrows : 65535u), dim3(256), 0, stream, rows, nrows,
                       nwords, stride_words, first_full_row);
    return hipGetLastError();
}

hipError_t launch_encode_text(const uint8_t *text, uint32_t n, uint64_t m, uint32_t nwords, uint32_t stride_words,
                              uint64_t *rows, unsigned long long *bad, hipStream_t stream)
{
    hipLaunchKernelGGL(encode_text_kernel, dim3((nwords + 255) / 256, n < 65535u ? n : 65535u), dim3(256), 0, stream, text, n, m, nwords,
                       stride_words, rows, bad);
    return hipGetLastError();
}

} // namespace lvbgpu
