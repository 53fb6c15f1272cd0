// propose_kernels.hip - NNI / SPR / TBR proposals AND their postorder programs generated on the
// device (SURVEY.md 8f "next" row 1: once scoring is fast, building candidates on the host and
// shipping them over PCIe is the limiter).
//
// One thread = one candidate.  It draws a move with the reference's rules
//   mutate_nni TreeOperations.c:160-209, mutate_spr :236-335, mutate_tbr :337-541
// (same eligibility/rejection loops, same re-use of the pruned parent as graft node; the random
// stream is a counter-based splitmix64, not the reference's generator), and writes
//   - the move as child-pair rewrites (edits) - what the host applies if the candidate is accepted,
//   - the token program fitch_walk will run (program.hpp's format), and its CandDesc.
//
// The dirty set of any of these moves is a union of at most two root-ward paths in the NEW
// topology (the reference's make_dirty_below calls), so the program is one or two chains, one
// merge, and the common path to the root; no general postorder is needed here (the general builder
// stays on the host: program.cpp).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdint.h>

#include "kernels.hpp"

namespace lvbgpu
{

namespace
{

struct DevRng
{
    uint64_t s;
    __device__ uint64_t next()
    {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    __device__ uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
};

// the emitting side of one candidate
struct Emit
{
    uint32_t *toks;
    int32_t *dsts;
    lvbgpu_edit_dev *edits;
    uint32_t ntok = 0, ndst = 0, nedit = 0, nfresh = 0, cap_t, cap_e;
    bool overflow = false;

    __device__ void tok(int32_t row, uint32_t flags)
    {
        if (ntok < cap_t)
            toks[ntok] = (uint32_t)row | flags;
        else
            overflow = true;
        ntok++;
        if (flags & TOK_FRESH)
            nfresh++;
    }
    __device__ void dst(int32_t d)
    {
        if (ndst < cap_t)
            dsts[ndst] = d;
        else
            overflow = true;
        ndst++;
    }
    __device__ void merge(int32_t d)
    {
        if (ntok > 0 && ntok <= cap_t)
            toks[ntok - 1] += 1u << TOK_MERGE_SHIFT;
        dst(d);
    }
    __device__ void edit(int32_t node, int32_t l, int32_t r)
    {
        if (nedit < cap_e)
            edits[nedit] = {node, l, r};
        else
            overflow = true;
        nedit++;
    }
};

struct Topo
{
    const int32_t *parent, *left, *right, *leaves;
    int32_t n, nb, root;
    __device__ int32_t sister(int32_t v) const
    {
        const int32_t p = parent[v];
        return left[p] == v ? right[p] : left[p];
    }
};

// children of v in the topology after an SPR-shaped move: pp lost sp for ss, dp lost dest for sp,
// sp holds (dest, top)
struct SprView
{
    const Topo &t;
    int32_t sp, ss, pp, dp, dest, top;
    __device__ void children(int32_t v, int32_t &l, int32_t &r) const
    {
        if (v == sp)
        {
            l = dest;
            r = top;
            return;
        }
        l = t.left[v];
        r = t.right[v];
        if (v == pp)
        {
            if (l == sp)
                l = ss;
            else
                r = ss;
        }
        if (v == dp)
        {
            if (l == dest)
                l = sp;
            else if (r == dest)
                r = sp;
        }
    }
    __device__ int32_t parent(int32_t v) const
    {
        if (v == sp)
            return dp;
        if (v == ss)
            return pp;
        if (v == dest || v == top)
            return sp;
        return t.parent[v];
    }
    __device__ int32_t other_child(int32_t v, int32_t d) const
    {
        int32_t l, r;
        children(v, l, r);
        return l == d ? r : l;
    }
};

constexpr int MAX_PATH = 768; // nodes of one root-ward path kept per thread; longer -> overflow flag

// One root-ward path of a thread.  In LDS (16-bit node numbers, element i of thread t at [i * 64 + t]: conflict-free)
// when the block's LDS holds it next to the topology, else in thread-private scratch memory, where every access is
// a trip to the cache hierarchy - the generator is a chain of dependent accesses and nothing else.
template <bool IN_LDS>
struct PathStore;
template <>
struct PathStore<true>
{
    uint16_t *p;
    int cap;
    __device__ int32_t get(int i) const { return (int32_t)p[(size_t)i * 64u]; }
    __device__ void set(int i, int32_t v) { p[(size_t)i * 64u] = (uint16_t)v; }
};
template <>
struct PathStore<false>
{
    int32_t buf[MAX_PATH];
    static constexpr int cap = MAX_PATH;
    __device__ int32_t get(int i) const { return buf[i]; }
    __device__ void set(int i, int32_t v) { buf[i] = v; }
};

} // namespace

// kind_all: 0 NNI, 1 SPR, 2 TBR; -1: candidate b gets kind b % 3; -2: NNI/SPR alternate by the
// parity of (mix_a + b) (reference -a 0, Solve.c:288-297); -3: drawn per candidate, NNI below
// threshold mix_a, SPR below mix_b, else TBR, both scaled to 2^32 (reference -a 1, Solve.c:262-283)
template <bool LDS_PATHS>
__global__ void propose_kernel(const int32_t *parent, const int32_t *left, const int32_t *right, const int32_t *leaves,
                               int32_t n, int32_t root, int32_t kind_all, uint32_t mix_a, uint32_t mix_b,
                               uint64_t seed, uint32_t B, uint32_t stride_t,
                               uint32_t stride_e, uint32_t *toks, int32_t *dsts, lvbgpu_edit_dev *edits,
                               CandDesc *cands, ProposalInfo *info, int32_t use_lds, const lvbgpu_move_dev *moves,
                               int32_t path_cap)
{
    // The walk below is pointer chasing (two root-ward paths, a random descent for TBR): from global
    // memory every step is an L2 round trip.  When the four arrays fit, the block first copies them
    // into LDS (coalesced) and chases there.
    extern __shared__ __attribute__((aligned(16))) int32_t lds_topo[];
    const int32_t nb_all = 2 * n - 3;
    if (use_lds)
    {
        // parent | left | right | leaves are contiguous and 16-byte aligned (launch_propose): 4 * nb ints = nb
        // 16-byte pieces.  Eight loads in flight per thread: one wave copies 16 KB, and a loop of single dwords
        // (62 dependent-looking round trips) was most of this kernel's time on small batches.
        const uint4 *src4 = reinterpret_cast<const uint4 *>(parent);
        uint4 *dst4 = reinterpret_cast<uint4 *>(lds_topo);
        for (int32_t i0 = 0; i0 < nb_all; i0 += 8 * (int32_t)blockDim.x)
        {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++)
            {
                const int32_t i = i0 + u * (int32_t)blockDim.x + (int32_t)threadIdx.x;
                if (i < nb_all)
                    v[u] = src4[i];
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
            {
                const int32_t i = i0 + u * (int32_t)blockDim.x + (int32_t)threadIdx.x;
                if (i < nb_all)
                    dst4[i] = v[u];
            }
        }
        __syncthreads();
        parent = lds_topo;
        left = lds_topo + nb_all;
        right = lds_topo + 2 * nb_all;
        leaves = lds_topo + 3 * nb_all;
    }
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B)
        return;
    const Topo t{parent, left, right, leaves, n, 2 * n - 3, root};
    DevRng rng{seed ^ ((uint64_t)(b + 1) * 0xD1B54A32D192ED03ull)};
    Emit e;
    e.toks = toks + (size_t)b * stride_t;
    e.dsts = dsts + (size_t)b * stride_t;
    e.edits = edits + (size_t)b * stride_e;
    e.cap_t = stride_t;
    e.cap_e = stride_e;
    int32_t kind = kind_all;
    if (kind_all == -1)
        kind = (int32_t)(b % 3u);
    else if (kind_all == -2)
        kind = ((mix_a + b) & 1u) ? 1 : 0;
    else if (kind_all == -3)
    {
        const uint32_t r = (uint32_t)(rng.next() >> 32);
        kind = r < mix_a ? 0 : (r < mix_b ? 1 : 2);
    }
    // moves != nullptr: nothing is drawn, candidate b IS moves[b] (validated by the host side of
    // lvbgpu_score_moves); everything after the draws is shared
    lvbgpu_move_dev given{0, -1, -1, -1};
    if (moves)
    {
        given = moves[b];
        kind = given.kind;
    }
    ProposalInfo pi{kind, -1, -1, -1, 0, 0, 0, 0};

    int32_t last = -1; // top node of the chain that reaches the root

    if (kind == 0)
    {
        // ---- NNI: u any internal node, v its parent, swap one child of u with u's sister
        const int32_t u = moves ? given.a : n + (int32_t)rng.below((uint32_t)(t.nb - n));
        const bool swap_right = moves ? given.b != 0 : (rng.next() >> 63) != 0;
        const int32_t v = t.parent[u], a = t.left[u], bb = t.right[u], c = t.sister(u);
        const int32_t keep = swap_right ? a : bb, moved = swap_right ? bb : a;
        pi.a = u;
        pi.flag = swap_right ? 1 : 0;
        // edits: v trades c for `moved`, u holds (keep, c)
        e.edit(v, t.left[v] == c ? moved : t.left[v], t.left[v] == c ? t.right[v] : moved);
        e.edit(u, keep, c);
        // chain: u (keep, c), then v with `moved` as its clean child, then the old path above v
        e.tok(keep, TOK_FRESH);
        e.tok(c, 0);
        e.dst(u);
        last = u;
        if (v != root)
        {
            e.tok(moved, 0);
            e.dst(v);
            last = v;
            for (int32_t w = t.parent[v]; w != root; w = t.parent[w])
            {
                e.tok(t.left[w] == last ? t.right[w] : t.left[w], 0);
                e.dst(w);
                last = w;
            }
            e.tok(t.left[root] == last ? t.right[root] : t.left[root], 0);
        }
        else
            e.tok(moved, 0); // the root's other child is now `moved`
        e.dst(-1);
        e.tok(root, 0);
        e.dst(-1);
    }
    else
    {
        // ---- SPR / TBR: prune src (with its parent sp), graft on the edge above dest
        // every draw loop is bounded: a tree with no admissible move must not hang the GPU
        int32_t src = given.a, dest = given.b;
        int tries = 0;
        if (!moves)
        {
            do
                src = (int32_t)rng.below((uint32_t)t.nb);
            while ((src == root || src == t.left[root] || src == t.right[root]) && ++tries < 4096);
            if (tries >= 4096)
            {
                e.overflow = true;
                src = t.left[t.left[root] >= n ? t.left[root] : t.right[root]];
            }
        }
        const int32_t sp = t.parent[src], ss = t.sister(src), pp = t.parent[sp];
        for (tries = moves ? 65536 : 0; tries < 65536; tries++)
        {
            dest = (int32_t)rng.below((uint32_t)t.nb);
            if (dest == src || dest == sp || dest == ss || dest == root)
                continue;
            bool below = false; // dest inside src's subtree?
            for (int32_t p = t.parent[dest]; p != UNSET; p = t.parent[p])
                if (p == src)
                {
                    below = true;
                    break;
                }
            if (!below)
                break;
        }
        if (!moves && tries >= 65536)
        {
            // no admissible destination found: emit nothing usable
            CandDesc none{};
            none.tok_off = b * stride_t;
            none.dst_off = b * stride_t;
            none.base = PROPOSAL_OVERFLOW_LENGTH;
            cands[b] = none;
            pi.overflow = 1;
            info[b] = pi;
            return;
        }
        const int32_t dp = t.parent[dest];
        pi.a = src;
        pi.b = dest;

        PathStore<LDS_PATHS> buf1, buf2; // root-ward paths
        if constexpr (LDS_PATHS)
        {
            // behind the topology copy: [2][path_cap][64 threads] node numbers
            uint16_t *paths = reinterpret_cast<uint16_t *>(lds_topo + 4 * nb_all);
            buf1.p = paths + threadIdx.x;
            buf1.cap = path_cap;
            buf2.p = paths + (size_t)path_cap * 64u + threadIdx.x;
            buf2.cap = path_cap;
        }
        const int cap = buf1.cap;
        int32_t top = src;   // what hangs under sp next to dest
        bool have_acc = false; // a chain inside the moved subtree already feeds sp
        if (kind == 2 && t.leaves[src] > 2 && !(moves && given.c < 0))
        {
            // TBR: re-root the moved subtree on the edge above a random leaf x (not a child of src)
            int32_t x = given.c;
            int xt = 0;
            if (!moves)
            {
                do
                {
                    x = src;
                    while (t.left[x] >= 0)
                    {
                        const int32_t l = t.left[x];
                        x = rng.below((uint32_t)t.leaves[x]) < (uint32_t)t.leaves[l] ? l : t.right[x];
                    }
                } while ((x == t.left[src] || x == t.right[src]) && ++xt < 4096);
            }
            if (xt >= 4096)
                e.overflow = true;
            pi.c = x;
            // path P0 = parent(x) .. Pk = src; walk it from the bottom to emit edits, then emit the
            // chain from Pk upwards: Pk (other child, displaced(k-1)), Pi (displaced(i-1)), P0 (x)
            PathStore<LDS_PATHS> &path = buf1;
            int k = 0;
            for (int32_t p = t.parent[x]; p != src && k < cap - 1; p = t.parent[p])
                path.set(k++, p);
            if (k >= cap - 1)
                e.overflow = true;
            path.set(k, src);
            // displaced[i] = the child of Pi that is not P(i-1) (for i = 0: the sister of x)
            // edits
            int32_t displaced = t.sister(x);
            e.edit(path.get(0), path.get(1), x);
            for (int i = 1; i < k; i++)
            {
                const int32_t pi_ = path.get(i);
                const int32_t other = (t.left[pi_] == path.get(i - 1)) ? t.right[pi_] : t.left[pi_];
                e.edit(pi_, path.get(i + 1), displaced);
                displaced = other;
            }
            {
                const int32_t l = t.left[src], r = t.right[src];
                const int32_t below_src = path.get(k - 1);
                e.edit(src, l == below_src ? displaced : l, l == below_src ? r : displaced);
                // chain bottom: src's two (clean) children in the new topology
                e.tok(l == below_src ? r : l, TOK_FRESH);
                e.tok(displaced, 0);
                e.dst(src);
            }
            // upwards: P(k-1) .. P1 each take the sister displaced from the node below them
            for (int i = k - 1; i >= 1; i--)
            {
                // clean child of Pi in the new topology = displaced(i-1) = child of P(i-1) not on the path
                const int32_t below_node = path.get(i - 1);
                const int32_t dis = (i - 1 == 0) ? t.sister(x)
                                                 : ((t.left[below_node] == path.get(i - 2)) ? t.right[below_node]
                                                                                            : t.left[below_node]);
                e.tok(dis, 0);
                e.dst(path.get(i));
            }
            e.tok(x, 0);
            e.dst(path.get(0));
            top = path.get(0);
            have_acc = true;
        }
        const SprView nv{t, sp, ss, pp, dp, dest, top};
        // edits of the prune-and-graft (pp and dp may be the same node)
        {
            int32_t l, r;
            if (pp == dp)
            {
                nv.children(pp, l, r);
                e.edit(pp, l, r);
            }
            else
            {
                nv.children(pp, l, r);
                e.edit(pp, l, r);
                nv.children(dp, l, r);
                e.edit(dp, l, r);
            }
            e.edit(sp, dest, top);
        }
        // path A: sp upwards in the new topology (excluding the root)
        PathStore<LDS_PATHS> &pa = buf1; // the TBR path above is no longer needed (top is saved)
        int na = 0;
        for (int32_t v = sp; v != root && na < cap; v = nv.parent(v))
            pa.set(na++, v);
        if (na >= cap)
            e.overflow = true;
        // path B: pp upwards until it meets A or the root.  Both paths end just below the root, so
        // what they share is a common suffix: walk B to the top, then strip the suffix (linear,
        // instead of searching A for every node of B)
        PathStore<LDS_PATHS> &pb = buf2;
        int nbp = 0, meet_a = -1;
        if (pp != root)
        {
            for (int32_t v = pp; v != root && nbp < cap; v = nv.parent(v))
                pb.set(nbp++, v);
            if (nbp >= cap)
                e.overflow = true;
            int common = 0;
            while (common < na && common < nbp && pa.get(na - 1 - common) == pb.get(nbp - 1 - common))
                common++;
            if (common > 0)
            {
                meet_a = na - common; // first node of A that B runs into
                nbp -= common;
            }
        }
        // chain below sp -> sp: sp's clean child is dest (its other child is `top`: clean for SPR,
        // the accumulated chain for TBR)
        auto chain_from_sp = [&](int upto) { // emits pa[0 .. upto)
            if (upto <= 0)
                return;
            if (have_acc)
                e.tok(dest, 0);
            else
            {
                e.tok(dest, TOK_FRESH);
                e.tok(top, 0);
            }
            e.dst(sp);
            for (int i = 1; i < upto; i++)
            {
                e.tok(nv.other_child(pa.get(i), pa.get(i - 1)), 0);
                e.dst(pa.get(i));
            }
        };
        if (nbp == 0)
        {
            // pp is the root, or pp already lies on A: one chain
            chain_from_sp(na);
            last = pa.get(na - 1);
        }
        else
        {
            // does B pass through sp (dest is pp or above it)?  then B continues as A: one chain from pp
            bool b_has_sp = (meet_a == 0);
            if (b_has_sp)
            {
                // B: pp (ss, other) ... up to the node below sp, then all of A with the chain as dirty child
                int32_t l, r;
                nv.children(pb.get(0), l, r);
                uint32_t push = have_acc ? TOK_PUSH : 0u;
                e.tok(l, TOK_FRESH | push);
                e.tok(r, 0);
                e.dst(pb.get(0));
                for (int i = 1; i < nbp; i++)
                {
                    e.tok(nv.other_child(pb.get(i), pb.get(i - 1)), 0);
                    e.dst(pb.get(i));
                }
                // sp: children (dest = top of B chain [dirty], top)
                if (have_acc)
                    e.merge(sp); // both children of sp are dirty: the subtree chain waits on the stack
                else
                {
                    e.tok(top, 0);
                    e.dst(sp);
                }
                for (int i = 1; i < na; i++)
                {
                    e.tok(nv.other_child(pa.get(i), pa.get(i - 1)), 0);
                    e.dst(pa.get(i));
                }
                last = pa.get(na - 1);
            }
            else
            {
                // two chains: A below the meeting node (or all of A if they only meet at the root), then B
                const int a_len = meet_a >= 0 ? meet_a : na;
                chain_from_sp(a_len);
                int32_t l, r;
                nv.children(pb.get(0), l, r);
                e.tok(l, TOK_FRESH | TOK_PUSH);
                e.tok(r, 0);
                e.dst(pb.get(0));
                for (int i = 1; i < nbp; i++)
                {
                    e.tok(nv.other_child(pb.get(i), pb.get(i - 1)), 0);
                    e.dst(pb.get(i));
                }
                if (meet_a >= 0)
                {
                    e.merge(pa.get(meet_a)); // both children of the meeting node are dirty
                    for (int i = meet_a + 1; i < na; i++)
                    {
                        e.tok(nv.other_child(pa.get(i), pa.get(i - 1)), 0);
                        e.dst(pa.get(i));
                    }
                    last = pa.get(na - 1);
                }
                else
                {
                    // the chains meet only at the root: its two children are both dirty
                    e.merge(-1);
                    last = -2; // root combine already emitted
                }
            }
        }
        if (last != -2)
        {
            e.tok(nv.other_child(root, last), 0);
            e.dst(-1);
        }
        e.tok(root, 0);
        e.dst(-1);
    }

    CandDesc cd{};
    cd.tok_off = b * stride_t;
    cd.ntok = e.ntok;
    cd.dst_off = b * stride_t;
    cd.ncomb = e.ndst;
    cd.base = 0;
    cd.flags = CAND_RESIDENT_BASE;
    cd.nfresh = e.nfresh;
    if (e.overflow)
    {
        // cannot be represented in the fixed strides: score it as the unchanged tree's root
        // combine only and flag it, so the host never accepts it
        cd.ntok = 0;
        cd.ncomb = 0;
        cd.nfresh = 0;
        cd.flags = 0;
        cd.base = PROPOSAL_OVERFLOW_LENGTH; // its "length": the host reads that as unusable
    }
    cands[b] = cd;
    pi.n_edits = (int32_t)e.nedit;
    pi.overflow = e.overflow ? 1 : 0;
    pi.ncomb = (int32_t)e.ndst;
    info[b] = pi;
}

hipError_t launch_propose(const int32_t *topo4, int32_t n, int32_t root, int32_t kind, uint32_t mix_a, uint32_t mix_b,
                          uint64_t seed, uint32_t B,
                          uint32_t stride_t, uint32_t stride_e, uint32_t *toks, int32_t *dsts, lvbgpu_edit_dev *edits,
                          CandDesc *cands, ProposalInfo *info, const lvbgpu_move_dev *moves, bool scratch_paths,
                          bool *paths_capped, hipStream_t stream)
{
    const int32_t nb = 2 * n - 3;
    // the dynamic-LDS ceiling is an attribute of the function ON A DEVICE: raise it once for each device a
    // context of this process launches on (the walk does the same per context, raise_lds_limit)
    static hipError_t raised_on[64];
    static bool asked_on[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipError_t raised = hipErrorInvalidDevice;
    if (dev >= 0 && dev < 64)
    {
        if (!asked_on[dev])
        {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&propose_kernel<true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(&propose_kernel<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            raised_on[dev] = e;
            asked_on[dev] = true;
        }
        raised = raised_on[dev];
    }
    const size_t lds_max = raised == hipSuccess ? 160u * 1024u : 64u * 1024u;
    // the topology in LDS when it fits (16 bytes per node; 160 KB of LDS per CU): up to ~5000 taxa
    size_t lds = (size_t)nb * 16u;
    int32_t use_lds = 1;
    if (lds > lds_max)
    {
        lds = 0;
        use_lds = 0;
    }
    // ... and behind it the threads' two root-ward paths: 256 bytes per path element and block.  No path has
    // more than n - 2 nodes; with less room than that a deep tree can overflow a candidate, which the caller
    // answers by running the batch again with the paths in scratch memory (*paths_capped says it may help)
    int32_t path_cap = 0;
    if (use_lds && !scratch_paths && nb <= 65535)
    {
        const size_t room = (lds_max - lds) / 256u;
        const size_t want = (size_t)std::min(n, MAX_PATH);
        path_cap = (int32_t)std::min(room, want);
        if (path_cap < 64 && path_cap < n - 1)
            path_cap = 0; // too little to be worth a second launch now and then
    }
    if (paths_capped)
        *paths_capped = path_cap > 0 && path_cap < std::min(n - 1, MAX_PATH);
    const dim3 grid((B + 63) / 64), block(64);
    if (path_cap > 0)
        hipLaunchKernelGGL(propose_kernel<true>, grid, block, lds + (size_t)path_cap * 256u, stream, topo4, topo4 + nb,
                           topo4 + 2 * nb, topo4 + 3 * nb, n, root, kind, mix_a, mix_b, seed, B, stride_t, stride_e, toks, dsts,
                           edits, cands, info, use_lds, moves, path_cap);
    else
        hipLaunchKernelGGL(propose_kernel<false>, grid, block, lds, stream, topo4, topo4 + nb, topo4 + 2 * nb,
                           topo4 + 3 * nb, n, root, kind, mix_a, mix_b, seed, B, stride_t, stride_e, toks, dsts, edits, cands,
                           info, use_lds, moves, 0);
    return hipGetLastError();
}

} // namespace lvbgpu
