// propose_kernels.hip - NNI / SPR / TBR proposals AND their postorder programs generated on the
// device (SURVEY.md 8f "next" row 1: once scoring is fast, building candidates on the host and
// shipping them over PCIe is the limiter).
//
// One WAVEFRONT = one candidate.  The wave draws a move with the reference's rules
//   mutate_nni TreeOperations.c:160-209, mutate_spr :236-335, mutate_tbr :337-541
// (same admissible sets, uniform over them as the reference's rejection loops are, same re-use of the
// pruned parent as graft node; the random stream is a counter-based splitmix64, not the reference's
// generator), and writes
//   - the move as child-pair rewrites (edits) - what the host applies if the candidate is accepted,
//   - the token program fitch_walk will run (program.hpp's format), and its CandDesc.
//
// Why a wave and not a thread.  The dirty set of these moves is a union of at most two root-ward paths
// in the NEW topology (the reference's make_dirty_below calls), so the program is one or two chains,
// one merge, and the common path to the root.  A thread walking those paths is ~5000 dependent
// instructions (46 us for any batch size, and milliseconds on a tree whose top is a caterpillar: the
// "is dest inside the pruned subtree" test walked to the root for every rejected draw).  Here nothing
// is walked.  With the topology the host uploads, per node, its depth, its preorder number, the leaves
// below it and its 2^k-th ancestors (binary lifting); then
//   * "v lies in the subtree of a" is one interval test on preorder numbers,
//   * the node at position j of a root-ward path is the j-th ancestor: lane j computes it in <= K table
//     reads, all positions at once,
//   * the clean sibling that position feeds on is "the child of that node whose subtree does not
//     hold the path's start": another interval test,
//   * rejection sampling draws 64 candidates at a time, one per lane, and takes the first admissible
//     one in draw order (a ballot),
// so a program is a handful of wave-wide pieces, and tokens, destinations and rewrites leave as coalesced
// stores.  The tables live in LDS (copied once per workgroup, ~32 KB at 500 taxa).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <stdint.h>

#include "kernels.hpp"
#include "gather.hpp"
#include "walk_body.hpp"

namespace lvbgpu
{

namespace
{

// counter-based random numbers: draw `idx` of stream `stream` of candidate `key` (splitmix64 finaliser)
__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct DevRng
{
    uint64_t key;
    __device__ uint64_t draw(uint32_t stream, uint32_t idx) const
    {
        return mix64(key + 0x9E3779B97F4A7C15ull * ((uint64_t)stream << 32 | (uint64_t)(idx + 1u)));
    }
    // uniform integer in [0, n)
    __device__ uint32_t below(uint32_t stream, uint32_t idx, uint32_t n) const
    {
        return (uint32_t)(((draw(stream, idx) >> 32) * (uint64_t)n) >> 32);
    }
};

// the tables of one topology (GenArgs in kernels.hpp gives the layout), in LDS or in global memory.  The pointers
// carry their address space: through generic pointers every table read is a FLAT load (the vector memory pipe decides
// at run time where the address lies: ~90 wait cycles each, 87 of them per candidate), with LDS pointers it is a
// ds_read (generator 19 -> 10 us at B = 4096).
template <typename IdxT, bool IN_LDS>
struct TabPtr
{
    using type = const __attribute__((address_space(1))) IdxT *;
};
template <typename IdxT>
struct TabPtr<IdxT, true>
{
    using type = const __attribute__((address_space(3))) IdxT *;
};
template <typename IdxT, bool IN_LDS>
struct Tab
{
    using P = typename TabPtr<IdxT, IN_LDS>::type;
    P parent, left, right, nleaf, depth, tin, first_leaf, leaf_order, up;
    int32_t n, nb, root, K;

    __device__ int32_t par(int32_t v) const { return (int32_t)parent[v]; }
    __device__ int32_t dep(int32_t v) const { return (int32_t)depth[v]; }
    // v in the subtree of a (a itself included); a is not the root
    __device__ bool inside(int32_t v, int32_t a) const
    {
        return (uint32_t)tin[v] - (uint32_t)tin[a] < 2u * (uint32_t)nleaf[a] - 1u;
    }
    __device__ int32_t other(int32_t v, int32_t c) const
    {
        const int32_t l = (int32_t)left[v];
        return l == c ? (int32_t)right[v] : l;
    }
    __device__ int32_t sister(int32_t v) const { return other(par(v), v); }
    // the child of v whose subtree does NOT hold y (y lies strictly below v)
    __device__ int32_t away(int32_t v, int32_t y) const
    {
        const int32_t l = (int32_t)left[v];
        return inside(y, l) ? (int32_t)right[v] : l;
    }
    // the child of v whose subtree holds y
    __device__ int32_t toward(int32_t v, int32_t y) const
    {
        const int32_t l = (int32_t)left[v];
        return inside(y, l) ? l : (int32_t)right[v];
    }
    // k-th ancestor (saturates at the root)
    __device__ int32_t anc(int32_t v, uint32_t k) const
    {
        for (int32_t i = 0; k != 0u && i < K; i++, k >>= 1)
            if (k & 1u)
                v = (int32_t)up[(size_t)i * nb + v];
        return k != 0u ? root : v;
    }
    // lowest common ancestor of two nodes below the root (may be the root)
    __device__ int32_t lca(int32_t a, int32_t b) const
    {
        int32_t da = dep(a), db = dep(b);
        if (da < db)
        {
            const int32_t s = a;
            a = b;
            b = s;
            const int32_t sd = da;
            da = db;
            db = sd;
        }
        a = anc(a, (uint32_t)(da - db));
        if (a == b)
            return a;
        // both are db below the root now: a jump of 2^i > db lands on the root for both, nothing to learn from it
        int32_t top = 31 - __builtin_clz((uint32_t)db | 1u);
        if (top > K - 1)
            top = K - 1;
        for (int32_t i = top; i >= 0; i--)
        {
            const int32_t ua = (int32_t)up[(size_t)i * nb + a], ub = (int32_t)up[(size_t)i * nb + b];
            if (ua != ub)
            {
                a = ua;
                b = ub;
            }
        }
        return par(a);
    }
};

typedef __attribute__((address_space(3))) uint32_t lds_word;

struct GenOut
{
    uint32_t *toks;
    int32_t *dsts;
    lvbgpu_edit_dev *edits;
    uint32_t ntok, ndst, nedit, nfresh, cap_e;
    lds_word *ltok; // pairing (GenArgs::pairs): the program's first 64 tokens go to this row of LDS as well (null: not)
};

template <typename IdxT, bool IN_LDS>
struct Gen
{
    const Tab<IdxT, IN_LDS> &t;
    GenOut o;
    uint32_t lane;

    // ---- pieces of a program (all arguments wave-uniform)
    // chain head with two clean children
    __device__ __forceinline__ void head2(int32_t row_a, uint32_t flags_a, int32_t row_b, int32_t dst, bool merge_after)
    {
        if (lane == 0)
        {
            const uint32_t ta = (uint32_t)row_a | TOK_FRESH | flags_a, tb = (uint32_t)row_b | (merge_after ? 1u << TOK_MERGE_SHIFT : 0u);
            o.toks[o.ntok] = ta;
            o.toks[o.ntok + 1] = tb;
            o.dsts[o.ndst] = dst;
            if (o.ltok && o.ntok + 1u < 64u)
            {
                o.ltok[o.ntok] = ta;
                o.ltok[o.ntok + 1] = tb;
            }
        }
        o.ntok += 2;
        o.ndst += 1;
        o.nfresh += 1;
    }
    // one node fed by the running set and one clean row
    __device__ __forceinline__ void one(int32_t row, int32_t dst, bool merge_after = false)
    {
        if (lane == 0)
        {
            const uint32_t tk = (uint32_t)row | (merge_after ? 1u << TOK_MERGE_SHIFT : 0u);
            o.toks[o.ntok] = tk;
            o.dsts[o.ndst] = dst;
            if (o.ltok && o.ntok < 64u)
                o.ltok[o.ntok] = tk;
        }
        o.ntok += 1;
        o.ndst += 1;
    }
    // the destination of a merge (its count rides on the token emitted just before: merge_after / merge_last)
    __device__ __forceinline__ void merge_dst(int32_t dst)
    {
        if (lane == 0)
            o.dsts[o.ndst] = dst;
        o.ndst += 1;
    }
    // `count` consecutive nodes of the OLD root-ward path of y, from its j0-th ancestor on, each fed by the
    // running set from below and by its clean child: the one given for position 0 of the path (row0, y's own
    // clean child in the new topology), otherwise the child that does not hold y - where a child that is the
    // pruned parent's old place (fix_from) now holds its sister (fix_to).  Lane-parallel, 64 nodes per trip.
    __device__ __forceinline__ void run(int32_t y, uint32_t j0, uint32_t count, int32_t row0, int32_t fix_from, int32_t fix_to, bool merge_last)
    {
        for (uint32_t base = 0; base < count; base += 64u)
        {
            const uint32_t idx = base + lane;
            if (idx < count)
            {
                const uint32_t j = j0 + idx;
                const int32_t v = t.anc(y, j);
                int32_t row = j == 0u ? row0 : t.away(v, y);
                if (row == fix_from)
                    row = fix_to;
                const uint32_t tk = (uint32_t)row | ((merge_last && idx == count - 1u) ? 1u << TOK_MERGE_SHIFT : 0u);
                o.toks[o.ntok + idx] = tk;
                o.dsts[o.ndst + idx] = v;
                if (o.ltok && o.ntok + idx < 64u)
                    o.ltok[o.ntok + idx] = tk;
            }
        }
        o.ntok += count;
        o.ndst += count;
    }
    __device__ __forceinline__ void edit(int32_t node, int32_t l, int32_t r)
    {
        if (lane == 0 && o.nedit < o.cap_e)
            o.edits[o.nedit] = {node, l, r};
        o.nedit += 1;
    }
    // the root's two combines: its clean child (new topology), then the root leaf's own row
    __device__ __forceinline__ void finish(int32_t root_clean_child)
    {
        if (root_clean_child >= 0)
            one(root_clean_child, -1);
        one(t.root, -1);
    }
};

// kind_all: 0 NNI, 1 SPR, 2 TBR; -1: candidate b gets kind b % 3; -2: NNI/SPR alternate by the
// parity of (mix_a + b) (reference -a 0, Solve.c:288-297); -3: drawn per candidate, NNI below
// threshold mix_a, SPR below mix_b, else TBR, both scaled to 2^32 (reference -a 1, Solve.c:262-283)
// bl: the candidate's index within its segment (what the draw is a function of, with the segment's seed);
// b = sg.start + bl: its slot in the batch
template <typename IdxT, bool IN_LDS>
__device__ void generate_one(const Tab<IdxT, IN_LDS> &t, const GenArgs &g, const GenSeg &sg, const uint32_t bl, const uint32_t lane,
                             lds_word *const pair_row = nullptr, lds_word *const pair_ntok = nullptr)
{
    const int32_t n = t.n, nb = t.nb, root = t.root;
    const uint32_t b = sg.start + bl;
    const uint64_t seed = ((uint64_t)sg.seed_hi << 32) | sg.seed_lo;
    const DevRng rng{seed ^ ((uint64_t)(bl + 1u) * 0xD1B54A32D192ED03ull)};
    Gen<IdxT, IN_LDS> e{t, GenOut{g.toks + (size_t)b * g.stride_t, g.dsts + (size_t)b * g.stride_t, g.edits + (size_t)b * g.stride_e, 0, 0,
                          0, 0, g.stride_e, pair_row},
                lane};
    // LVBGPU_GEN_PROFILE (tools/gen_profile.py): where a candidate's time goes, clock stamps of the first 256 candidates
    auto stamp = [&](uint32_t k) {
        if (g.prof && b < 256u && lane == 0)
            g.prof[b * 8u + k] = __builtin_readcyclecounter();
    };
    stamp(0);
    int32_t kind = sg.kind_all;
    if (sg.kind_all == -1)
        kind = (int32_t)(bl % 3u);
    else if (sg.kind_all == -2)
        kind = ((sg.mix_a + bl) & 1u) ? 1 : 0;
    else if (sg.kind_all == -3)
    {
        const uint32_t r = (uint32_t)(rng.draw(0, 0) >> 32);
        kind = r < sg.mix_a ? 0 : (r < sg.mix_b ? 1 : 2);
    }
    // moves != nullptr: nothing is drawn, candidate b IS moves[b] (validated by the host side of
    // lvbgpu_score_moves); everything after the draws is shared
    // (as scalars: a struct assigned under a condition lived in scratch memory)
    int32_t given_a = -1, given_b = -1, given_c = -1;
    if (g.moves)
    {
        const lvbgpu_move_dev mv = g.moves[b];
        kind = mv.kind;
        given_a = mv.a;
        given_b = mv.b;
        given_c = mv.c;
    }
    // (the move's description as scalars: as a struct filled field by field it lived in scratch memory)
    int32_t pi_a = -1, pi_b = -1, pi_c = -1, pi_flag = 0;
    bool unusable = false;

    if (kind == 0)
    {
        // ---- NNI: u any internal node, v its parent, swap one child of u with u's sister
        const int32_t u = g.moves ? given_a : n + (int32_t)rng.below(1, 0, (uint32_t)(nb - n));
        const bool swap_right = g.moves ? given_b != 0 : (rng.draw(1, 1) >> 63) != 0;
        const int32_t v = t.par(u), a = (int32_t)t.left[u], bb = (int32_t)t.right[u], c = t.sister(u);
        const int32_t keep = swap_right ? a : bb, moved = swap_right ? bb : a;
        pi_a = u;
        pi_flag = swap_right ? 1 : 0;
        // edits: v trades c for `moved` (same side), u holds (keep, c)
        {
            const int32_t vl = (int32_t)t.left[v], vr = (int32_t)t.right[v];
            e.edit(v, vl == c ? moved : vl, vl == c ? vr : moved);
            e.edit(u, keep, c);
        }
        // chain: u (keep, c), then v with `moved` as its clean child, then the old path above v
        e.head2(keep, 0u, c, u, false);
        if (v != root)
        {
            const uint32_t dv = (uint32_t)t.dep(v);
            e.run(v, 0u, dv, moved, -1, -1, false);
            e.finish(t.away(root, v));
        }
        else
            e.finish(moved); // the root's other child is now `moved`
    }
    else
    {
        // ---- SPR / TBR: prune src (with its parent sp), graft on the edge above dest
        int32_t src = given_a, dest = given_b;
        if (!g.moves)
        {
            // src: any node but the root and its two children (TreeOperations.c:256-259)
            const int32_t r1 = (int32_t)t.left[root], r2 = (int32_t)t.right[root];
            src = -1;
            for (uint32_t round = 0; round < 16u && src < 0; round++)
            {
                const int32_t cand = (int32_t)rng.below(2, round * 64u + lane, (uint32_t)nb);
                const uint64_t ok = __builtin_amdgcn_ballot_w64(cand != root && cand != r1 && cand != r2);
                if (ok)
                    src = __builtin_amdgcn_readlane(cand, (int)__builtin_ctzll(ok));
            }
        }
        int32_t sp = -1, ss = -1, pp = -1;
        if (src >= 0)
        {
            sp = t.par(src);
            ss = t.other(sp, src);
            pp = t.par(sp);
        }
        if (!g.moves && src >= 0)
        {
            // dest: any node but src, its parent, its sister, the root, and src's own subtree (268-271 and the
            // descendant test); 64 draws per round, the first admissible one in draw order wins
            dest = -1;
            for (uint32_t round = 0; round < 1024u && dest < 0; round++)
            {
                const int32_t cand = (int32_t)rng.below(3, round * 64u + lane, (uint32_t)nb);
                const bool good = cand != sp && cand != ss && cand != root && !t.inside(cand, src);
                const uint64_t ok = __builtin_amdgcn_ballot_w64(good);
                if (ok)
                    dest = __builtin_amdgcn_readlane(cand, (int)__builtin_ctzll(ok));
            }
        }
        if (src < 0 || dest < 0)
            unusable = true; // no admissible move found: emit nothing usable
        else
        {
            stamp(1); // drawn
            const int32_t dp = t.par(dest);
            pi_a = src;
            pi_b = dest;

            int32_t top = src;     // what hangs under sp next to dest
            bool have_acc = false; // a chain inside the moved subtree already feeds sp
            if (kind == 2 && (int32_t)t.nleaf[src] > 2 && !(g.moves && given_c < 0))
            {
                // TBR: re-root the moved subtree on the edge above a random leaf x (not a child of src)
                int32_t x = given_c;
                if (!g.moves)
                {
                    const int32_t c1 = (int32_t)t.left[src], c2 = (int32_t)t.right[src];
                    const uint32_t first = (uint32_t)t.first_leaf[src], cnt = (uint32_t)t.nleaf[src];
                    x = -1;
                    for (uint32_t round = 0; round < 64u && x < 0; round++)
                    {
                        const int32_t cand = (int32_t)t.leaf_order[first + rng.below(4, round * 64u + lane, cnt)];
                        const uint64_t ok = __builtin_amdgcn_ballot_w64(cand != c1 && cand != c2);
                        if (ok)
                            x = __builtin_amdgcn_readlane(cand, (int)__builtin_ctzll(ok));
                    }
                }
                if (x < 0)
                    unusable = true;
                else
                {
                    pi_c = x;
                    // path P0 = parent(x) .. Pk = src (P_i = the (i+1)-th ancestor of x); k >= 1
                    const uint32_t k = (uint32_t)(t.dep(x) - t.dep(src)) - 1u;
                    const int32_t p0 = t.par(x), sis_x = t.other(p0, x);
                    const int32_t below_src = t.toward(src, x); // P(k-1)
                    const int32_t displaced_k = k == 1u ? sis_x : t.away(below_src, x);
                    // rewrites: P0 = (P1, x); P_i = (P_(i+1), displaced_i), displaced_1 = sister(x), displaced_i = the
                    // child of P_(i-1) off the path; src trades P(k-1) for the last displaced one (same side)
                    if (lane == 0 && e.o.nedit < e.o.cap_e)
                        e.o.edits[e.o.nedit] = {p0, t.par(p0), x};
                    for (uint32_t base = 1; base < k; base += 64u)
                    {
                        const uint32_t i = base + lane;
                        if (i < k && e.o.nedit + i < e.o.cap_e)
                        {
                            const int32_t node = t.anc(x, i + 1u);
                            const int32_t disp = i == 1u ? sis_x : t.away(t.toward(node, x) /* P(i-1) */, x);
                            e.o.edits[e.o.nedit + i] = {node, t.par(node), disp};
                        }
                    }
                    e.o.nedit += k;
                    {
                        const int32_t l = (int32_t)t.left[src], r = (int32_t)t.right[src];
                        e.edit(src, l == below_src ? displaced_k : l, l == below_src ? r : displaced_k);
                        // chain bottom: src's two (clean) children in the new topology
                        e.head2(l == below_src ? r : l, 0u, displaced_k, src, false);
                    }
                    // upwards in the NEW subtree: P(k-1) .. P1, each fed by the sister displaced from the node below it
                    for (uint32_t base = 0; base + 1u < k; base += 64u)
                    {
                        const uint32_t q = base + lane;
                        if (q + 1u < k)
                        {
                            const uint32_t i = k - 1u - q; // P_i, i = k-1 .. 1
                            const int32_t node = t.anc(x, i + 1u);
                            const int32_t dis = i == 1u ? sis_x : t.away(t.toward(node, x), x);
                            e.o.toks[e.o.ntok + q] = (uint32_t)dis;
                            e.o.dsts[e.o.ndst + q] = node;
                            if (e.o.ltok && e.o.ntok + q < 64u)
                                e.o.ltok[e.o.ntok + q] = (uint32_t)dis;
                        }
                    }
                    e.o.ntok += k - 1u;
                    e.o.ndst += k - 1u;
                    e.one(x, p0);
                    top = p0;
                    have_acc = true;
                }
            }
            if (!unusable)
            {
                // rewrites of the prune-and-graft (pp and dp may be the same node): pp trades sp for ss, dp trades dest
                // for sp, each on the side it was; sp holds (dest, top)
                {
                    int32_t l = (int32_t)t.left[pp], r = (int32_t)t.right[pp];
                    if (l == sp)
                        l = ss;
                    else
                        r = ss;
                    if (pp == dp)
                    {
                        if (l == dest)
                            l = sp;
                        else if (r == dest)
                            r = sp;
                        e.edit(pp, l, r);
                    }
                    else
                    {
                        e.edit(pp, l, r);
                        l = (int32_t)t.left[dp];
                        r = (int32_t)t.right[dp];
                        if (l == dest)
                            l = sp;
                        else
                            r = sp;
                        e.edit(dp, l, r);
                    }
                    e.edit(sp, dest, top);
                }
                stamp(2); // rewrites (and the TBR subtree chain) out
                // ---- the program.  d(v) = depth below the root; a root-ward path of y has d(y) nodes (the root is
                // not one of them).  Three shapes (TreeOperations.c:302, 330-334: both root-ward paths are dirty):
                const int32_t oc_dp = dp == root ? -1 : t.other(dp, dest); // dp's clean child once sp has taken dest's place
                const int32_t oc_pp = t.other(pp, sp);                     // pp's other child (pp keeps it; sp's place goes to ss)
                // the chain's head at sp: fed by dest and, if the moved subtree was recomputed, by the running set, else by its top
#define LVB_HEAD_SP()                                                                                                   \
    do                                                                                                                 \
    {                                                                                                                  \
        if (have_acc)                                                                                                  \
            e.one(dest, sp, false);                                                                                    \
        else                                                                                                           \
            e.head2(dest, 0u, top, sp, false);                                                                         \
    } while (0)
                if (dest != ss && t.inside(dest, ss))
                {
                    // (1) dest below the sister: one chain sp, dp .. ss, then pp .. (sp's old place is skipped)
                    LVB_HEAD_SP();
                    e.run(dp, 0u, (uint32_t)(t.dep(dp) - t.dep(ss)) + 1u, oc_dp, -1, -1, false);
                    if (pp != root)
                    {
                        e.run(pp, 0u, (uint32_t)t.dep(pp), oc_pp, -1, -1, false);
                        e.finish(t.away(root, pp));
                    }
                    else
                        e.finish(oc_pp);
                }
                else if (pp != root && t.inside(pp, dest))
                {
                    // (2) dest is pp or above it: one chain from pp through dest, sp (new place), dp ..
                    const uint32_t up_to_dest = (uint32_t)(t.dep(pp) - t.dep(dest)); // nodes above pp up to dest
                    e.head2(ss, have_acc ? TOK_PUSH : 0u, oc_pp, pp, have_acc && up_to_dest == 0u);
                    e.run(pp, 1u, up_to_dest, -1, -1, -1, have_acc);
                    if (have_acc)
                        e.merge_dst(sp); // both children of sp are dirty: the subtree chain waited on the stack
                    else
                        e.one(top, sp);
                    if (dp != root)
                    {
                        e.run(dp, 0u, (uint32_t)t.dep(dp), oc_dp, -1, -1, false);
                        e.finish(t.away(root, dp));
                    }
                    else
                        e.finish(t.other(root, dest));
                }
                else
                {
                    // (3) dest elsewhere: paths from sp (through dp) and from pp, meeting at M (possibly only at the root)
                    const int32_t m = (pp == root || dp == root) ? root : t.lca(dp, pp);
                    if (pp == root || m == pp)
                    {
                        // pp is the root or lies on the path from dp: one chain; where it passes pp, sp's place holds ss
                        LVB_HEAD_SP();
                        if (dp != root)
                        {
                            e.run(dp, 0u, (uint32_t)t.dep(dp), oc_dp, sp, ss, false);
                            e.finish(pp == root ? ss : t.away(root, dp));
                        }
                        else
                            e.finish(ss); // root = (ss, sp) now
                    }
                    else
                    {
                        const uint32_t dm = m == root ? 0u : (uint32_t)t.dep(m);
                        const uint32_t len_a = dp == root ? 0u : (uint32_t)t.dep(dp) - dm; // dp .. below M (0: M is dp)
                        const uint32_t len_b = (uint32_t)t.dep(pp) - dm;                     // pp .. below M (>= 1)
                        LVB_HEAD_SP();
                        e.run(dp, 0u, len_a, oc_dp, -1, -1, false);
                        e.head2(ss, TOK_PUSH, oc_pp, pp, len_b == 1u);
                        e.run(pp, 1u, len_b - 1u, -1, -1, -1, true);
                        e.merge_dst(m == root ? -1 : m); // both children of M are dirty
                        if (m != root)
                        {
                            e.run(m, 1u, dm - 1u, -1, -1, -1, false);
                            e.finish(t.away(root, m));
                        }
                        else
                            e.finish(-1); // the chains met at the root: its children combine was the merge
                    }
                }
            }
        }
    }

    stamp(3); // program out
    const bool overflow = unusable || e.o.nedit > e.o.cap_e;
    if (lane == 0)
    {
        CandDesc cd{};
        cd.tok_off = b * g.stride_t;
        cd.dst_off = b * g.stride_t;
        if (overflow)
        {
            // cannot be represented (no admissible move, or a TBR path longer than the rewrite stride): its
            // "length" tells the host never to accept it
            cd.base = PROPOSAL_OVERFLOW_LENGTH;
        }
        else
        {
            cd.ntok = e.o.ntok;
            cd.ncomb = e.o.ndst;
            cd.flags = CAND_RESIDENT_BASE | ((uint32_t)sg.chain << CAND_CHAIN_SHIFT);
            cd.nfresh = e.o.nfresh;
        }
        g.cands[b] = cd;
        g.len_out[b] = 0ull; // the walk accumulates into it: cleared here, so a step needs no clearing pass of its own
        g.info[b] = ProposalInfo{kind, pi_a, pi_b, pi_c, pi_flag, (int32_t)e.o.nedit, overflow ? 1 : 0, (int32_t)e.o.ndst};
        if (pair_ntok) // how much of the program lies in LDS for the workgroup's pairing: all of it, or nothing to share
            *pair_ntok = (overflow || e.o.ntok > 64u) ? 0u : e.o.ntok;
    }
    stamp(4);
}

} // namespace

#undef LVB_HEAD_SP
// Who walks with whom (fitch_walk_pair; GenArgs::pairs): the sixteen candidates a generating workgroup has just drawn are
// paired among themselves, by the length of their programs' common END - the number of row loads a wave saves by walking
// the two together - longest first (greedy; ties: the pair of the later candidates; a function of the draw, nothing of
// timing).  Every wave left its program's tokens in a row of LDS.  Wave w compares its own, turned round, a token per
// lane, with those of the eight candidates behind it (cyclically: every pair exactly once, eight rows in flight at a time);
// then sixteen lanes of one wave - lane w is candidate w - repeat: everybody names the best partner still free, pairs
// that name each other are made (with a strict order on the pairs that IS the greedy matching, in 3-4 rounds instead of
// eight searches for a maximum).  Random neighbours of a 500-taxon tree share this way (tools/shared_suffix_estimate.py,
// tools/pair_quality.py): NNI 0.24-0.35 of their row reads, SPR / TBR 0.11 at D = 20 and 0.20 at D = 52; a sort of the
// whole batch by one key - rounds 3 and 4, a role of its own at the launch's end that waited for every generating
// workgroup: 8-9 us - gave SPR 0.13 and 0.17.  first: the batch index of the iteration's first candidate; cnt: how many
// were drawn (1 .. 16); pair0: their first pair.  Called by all sixteen waves.
__device__ __forceinline__ void pair_the_drawn(const GenArgs &g, lds_word *const pl, const uint32_t first, const uint32_t cnt, const uint32_t pair0)
{
    lds_word *const tok = pl, *const ntok = pl + 16u * PAIR_ROW, *const share = ntok + 16u;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __syncthreads(); // the programs are out
    {
        const uint32_t nall = lane < cnt ? ntok[lane] : 0u; // (0: too long for one chunk, not a proposal, or not drawn: shares nothing)
        const uint32_t nw = (uint32_t)__builtin_amdgcn_readlane((int)nall, (int)wave);
        const uint32_t mine = lane < nw ? tok[wave * PAIR_ROW + nw - 1u - lane] : 0xFFFFFFFEu;
        uint32_t other[8];
#pragma unroll
        for (uint32_t d = 1; d <= 8u; d++)
        {
            const uint32_t j = (wave + d) & 15u;
            const uint32_t nj = (uint32_t)__builtin_amdgcn_readlane((int)nall, (int)j);
            other[d - 1u] = lane < nj ? tok[j * PAIR_ROW + nj - 1u - lane] : 0xFFFFFFFDu;
        }
#pragma unroll
        for (uint32_t d = 1; d <= 8u; d++)
        {
            const uint32_t j = (wave + d) & 15u;
            const uint32_t nj = (uint32_t)__builtin_amdgcn_readlane((int)nall, (int)j);
            const uint64_t differ = ~__builtin_amdgcn_ballot_w64(mine == other[d - 1u]);
            uint32_t k = differ ? (uint32_t)__builtin_ctzll(differ) : 64u;
            const uint32_t m = nw < nj ? nw : nj;
            k = k < m ? k : m;
            if (lane == 0 && wave < cnt && j < cnt && (d < 8u || wave < 8u))
            {
                const uint32_t lo = wave < j ? wave : j, hi = wave < j ? j : wave;
                const uint32_t v = ((k + 1u) << 8) | (lo << 4) | hi; // (never 0; no two pairs alike)
                share[wave * 16u + j] = v;
                share[j * 16u + wave] = v;
            }
        }
    }
    __syncthreads();
    if (wave != 0)
        return; // (the next iteration's rows and lengths are other words than `share`, which all write after the next barrier)
    uint32_t row[16];
#pragma unroll
    for (uint32_t j = 0; j < 16u; j++)
        row[j] = (lane < cnt && j < cnt && j != lane) ? share[(lane & 15u) * 16u + j] : 0u;
    uint32_t used = 0u, out = 0u;
    for (uint32_t round = 0; round < 8u; round++)
    {
        uint32_t best = 0u;
#pragma unroll
        for (uint32_t j = 0; j < 16u; j++)
            if (!((used >> j) & 1u) && row[j] > best)
                best = row[j];
        if (lane >= cnt || ((used >> (lane & 31u)) & 1u))
            best = 0u;
        const uint32_t e = best & 255u;
        const uint32_t partner = (e >> 4) == lane ? (e & 15u) : (e >> 4);
        const uint32_t theirs = (uint32_t)__shfl((int)best, (int)partner);
        const bool made = best != 0u && theirs == best; // both named this pair
        const uint32_t mask = (uint32_t)__builtin_amdgcn_ballot_w64(made);
        if (!mask)
            break;
        const bool writes = made && lane < partner;
        const uint32_t wm = (uint32_t)__builtin_amdgcn_ballot_w64(writes);
        if (writes)
        {
            const uint32_t slot = pair0 + out + (uint32_t)__builtin_popcount(wm & ((1u << lane) - 1u));
            g.pairs[2u * slot] = first + lane;
            g.pairs[2u * slot + 1u] = first + partner;
        }
        out += (uint32_t)__builtin_popcount(wm);
        used |= mask;
    }
    const uint32_t rest = ~used & ((1u << cnt) - 1u); // (an odd run's last one)
    if (rest && lane == 0)
    {
        g.pairs[2u * (pair0 + out)] = first + (uint32_t)__builtin_ctz(rest);
        g.pairs[2u * (pair0 + out) + 1u] = PICK_NONE;
    }
}

// The generator's part of a launch: workgroup `blk` of the `nblk` that generate (propose_kernel: the whole grid; the post
// launch: its last workgroups).  lds_tables: the workgroup's dynamic LDS.
template <typename IdxT, bool IN_LDS>
__device__ __forceinline__ void gen_role(const GenArgs &g, const uint32_t blk, const uint32_t nblk, uint4 *const lds_tables)
{
    const uint64_t t_enter = g.prof ? __builtin_readcyclecounter() : 0ull;
    // which segment (resident tree) this workgroup works for: its first workgroup is seg[s].blk_start
    uint32_t s = 0;
    while (s + 1u < g.nseg && blk >= g.seg[s + 1u].blk_start)
        s++;
    const GenSeg &sg = g.seg[s];
    const uint32_t blk_local = blk - sg.blk_start;
    const uint32_t seg_blocks = (s + 1u < g.nseg ? g.seg[s + 1u].blk_start : nblk) - sg.blk_start;
    const char *tables = reinterpret_cast<const char *>(g.tables) + (size_t)sg.chain * g.table_stride;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nthreads = blockDim.x, nwaves = nthreads >> 6;
    using P = typename TabPtr<IdxT, IN_LDS>::type;
    P tab;
    if constexpr (IN_LDS)
    {
        if (sg.wait)
        {
            // post launch: this tree's tables are being rebuilt by an earlier workgroup of the SAME launch (workgroups are
            // dealt in order: it is resident or done) - wait for its word, then copy with agent-scope loads: the tables
            // were written through, and a plain load could be served from a line this XCD's L2 still holds from the last step
            // (every wave polls for itself and they agree through a word of the DYNAMIC LDS, which the tables overwrite
            // afterwards: with any static LDS in this kernel - __syncthreads_or has some - the dynamic-LDS ceiling of
            // 160 KiB cannot be asked for)
            volatile uint32_t *const agree = reinterpret_cast<volatile uint32_t *>(lds_tables);
            if (threadIdx.x == 0)
                *agree = 0u;
            __syncthreads();
            uint32_t budget = g.wait_spins ? g.wait_spins : 1u << 24; // ~ seconds: a backstop, the rebuild finishes on its own
            if (lane == 0)
            {
                while (__hip_atomic_load(g.table_ready + sg.chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != g.ready_seq && --budget)
                    __builtin_amdgcn_s_sleep(16);
                if (budget == 0u)
                    *agree = 1u;
            }
            __syncthreads();
            const uint32_t gave_up = *agree;
            __syncthreads(); // (read by all before the copy below overwrites it)
            if (gave_up)
            {
                // never walk programs made from torn tables: these candidates are "not proposals"
                for (uint32_t bl = blk_local * nwaves + wave; bl < sg.count; bl += seg_blocks * nwaves)
                    if (lane == 0)
                    {
                        CandDesc cd{};
                        cd.tok_off = cd.dst_off = (sg.start + bl) * g.stride_t;
                        cd.base = PROPOSAL_OVERFLOW_LENGTH;
                        g.cands[sg.start + bl] = cd;
                        g.len_out[sg.start + bl] = 0ull;
                        g.info[sg.start + bl] = ProposalInfo{0, -1, -1, -1, 0, 0, 1, 0};
                        if (g.pairs && !(bl & 1u)) // (the paired walk still finds every candidate once: neighbours, as they come)
                        {
                            uint32_t pair_seg = 0;
                            for (uint32_t i = 0; i < s; i++)
                                pair_seg += (g.seg[i].count + 1u) >> 1;
                            g.pairs[2u * (pair_seg + (bl >> 1))] = sg.start + bl;
                            g.pairs[2u * (pair_seg + (bl >> 1)) + 1u] = bl + 1u < sg.count ? sg.start + bl + 1u : PICK_NONE;
                        }
                    }
                return;
            }
            const unsigned long long *src8 = reinterpret_cast<const unsigned long long *>(tables);
            unsigned long long *dst8 = reinterpret_cast<unsigned long long *>(lds_tables);
            const uint32_t n8 = g.table_bytes / 8u;
            for (uint32_t i = threadIdx.x; i < n8; i += nthreads)
                dst8[i] = __hip_atomic_load(src8 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        else
        {
            // one coalesced copy per workgroup; everything after it is LDS latency instead of L2 latency (1024 threads x 4
            // loads of 16 bytes in flight: the 34 KB of a 500-taxon tree are one round trip)
            const uint4 *src4 = reinterpret_cast<const uint4 *>(tables);
            const uint32_t n16 = g.table_bytes / 16u;
            constexpr uint32_t DEPTH = 4;
            for (uint32_t i0 = 0; i0 < n16; i0 += DEPTH * nthreads)
            {
                uint4 v[DEPTH];
#pragma unroll
                for (uint32_t u = 0; u < DEPTH; u++)
                {
                    const uint32_t i = i0 + u * nthreads + threadIdx.x;
                    v[u] = src4[i < n16 ? i : n16 - 1u]; // (unconditional: a conditionally filled array went through scratch)
                }
#pragma unroll
                for (uint32_t u = 0; u < DEPTH; u++)
                {
                    const uint32_t i = i0 + u * nthreads + threadIdx.x;
                    if (i < n16)
                        lds_tables[i] = v[u];
                }
            }
        }
        __syncthreads();
        tab = (P)lds_tables; // C-style: a generic pointer known to lie in LDS becomes an LDS pointer
    }
    else
        tab = (P)tables;
    const size_t nb = (size_t)g.nb;
    Tab<IdxT, IN_LDS> t;
    t.parent = tab;
    t.left = tab + nb;
    t.right = tab + 2 * nb;
    t.nleaf = tab + 3 * nb;
    t.depth = tab + 4 * nb;
    t.tin = tab + 5 * nb;
    t.first_leaf = tab + 6 * nb;
    t.leaf_order = tab + 7 * nb;
    t.up = tab + 7 * nb + (size_t)g.leaf_order_len;
    t.n = g.n;
    t.nb = g.nb;
    t.root = sg.root;
    t.K = g.K;
    if (g.prof && lane == 0 && sg.start + blk_local * nwaves + wave < 256u)
    {
        g.prof[(sg.start + blk_local * nwaves + wave) * 8u + 5u] = t_enter;
        g.prof[(sg.start + blk_local * nwaves + wave) * 8u + 6u] = __builtin_readcyclecounter();
    }
    if (!g.pairs)
    {
        for (uint32_t bl = blk_local * nwaves + wave; bl < sg.count; bl += seg_blocks * nwaves)
            generate_one(t, g, sg, bl, lane);
        return;
    }
    // two candidates per wave: sixteen candidates at a time, drawn and then paired (the workgroup in step)
    uint32_t pair_seg = 0; // the segment's first pair: behind those of the segments before it
    for (uint32_t i = 0; i < s; i++)
        pair_seg += (g.seg[i].count + 1u) >> 1;
    lds_word *const pl = (lds_word *)(reinterpret_cast<uint32_t *>(lds_tables) + (IN_LDS ? ((g.table_bytes + 15u) & ~15u) / 4u : 0u));
    for (uint32_t bl0 = blk_local * nwaves; bl0 < sg.count; bl0 += seg_blocks * nwaves)
    {
        if (bl0 + wave < sg.count)
            generate_one(t, g, sg, bl0 + wave, lane, pl + wave * PAIR_ROW, pl + 16u * PAIR_ROW + wave);
        const uint32_t cnt = sg.count - bl0 < nwaves ? sg.count - bl0 : nwaves;
        pair_the_drawn(g, pl, sg.start + bl0, cnt, pair_seg + (bl0 >> 1));
        if (g.prof && lane == 0 && sg.start + bl0 + wave < 256u)
            g.prof[(sg.start + bl0 + wave) * 8u + 7u] = __builtin_readcyclecounter(); // (paired)
    }
}

template <typename IdxT, bool IN_LDS>
__global__ __launch_bounds__(GEN_THREADS) void propose_kernel(const GenArgs g)
{
    extern __shared__ uint4 lds_dyn[];
    gen_role<IdxT, IN_LDS>(g, blockIdx.x, g.n_gen_blocks, lds_dyn);
}

// ---------------------------------------------------------------------------------------------
// The tables of a tree after an accepted device move, rebuilt ON the device (lvbgpu_chains_commit): one workgroup
// per picked chain applies the candidate's rewrites to left / right in that chain's table slot and derives everything
// else again - parents; leaves below, by letting every leaf count itself into its ancestors; depth, 2^k-th
// ancestors, preorder number and first-leaf position from ONE walk of each node along its own root-ward path (no
// level-by-level rounds: a barrier of 16 waves costs as much as a path).  No host work and no
// upload per accepted move; with R chains accepting in one step that was most of the step.
// With ga.k != 0 wave 0 of the workgroup of pick j first sends that pick's descriptor and rewrites to the host
// (gather.hpp: done in a few microseconds, so the host has its flag long before the rebuilds end).
// g.table_ready != null (post launch): the generator of the SAME launch waits for these tables - they are written
// through to where every XCD sees them (agent-scope stores), and the chain's word says when.
template <typename T>
__device__ __forceinline__ void store_tab(T *p, T v, bool through)
{
    if (through)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *p = v;
}

template <typename IdxT>
__device__ __forceinline__ void rebuild_role(const RebuildArgs &g, const GatherArgs &ga, const GenArgs &gen, const uint32_t j,
                                             int32_t *const lds_i32)
{
    const bool is_ext = j >= g.n_pick;
    const uint32_t x = j - g.n_pick; // (is_ext)
    const uint32_t cand = is_ext ? 0u : g.pick_idx[j];
    if (!is_ext && j < ga.k && threadIdx.x < 64u)
        gather_one_pick(ga, j, threadIdx.x);
    if (!is_ext && !g.rebuild_picks)
        return; // (uniform for the workgroup: no barrier is skipped by some)
    const uint32_t chain = is_ext ? (uint32_t)g.ext[x].chain : g.cands[cand].flags >> CAND_CHAIN_SHIFT;
    const bool through = g.table_ready != nullptr && ((g.wait_mask >> chain) & 1ull) != 0ull;
    IdxT *tab = reinterpret_cast<IdxT *>(reinterpret_cast<char *>(g.tables) + (size_t)chain * g.table_stride);
    const int32_t n = g.n, nb = g.nb, K = g.K;
    IdxT *t_parent = tab, *t_left = tab + nb, *t_right = tab + 2 * (size_t)nb, *t_nleaf = tab + 3 * (size_t)nb,
         *t_depth = tab + 4 * (size_t)nb, *t_tin = tab + 5 * (size_t)nb, *t_first = tab + 6 * (size_t)nb,
         *t_order = tab + 7 * (size_t)nb, *t_up = tab + 7 * (size_t)nb + g.leaf_order_len;
    int32_t *left = lds_i32, *right = left + nb, *parent = right + nb, *nleaf = parent + nb, *shared = nleaf + nb; // shared[0] root
    // one step of a root-ward walk as ONE 8-byte LDS read: {parent, weight}; 8-byte aligned behind the int32 arrays
    uint2 *step = reinterpret_cast<uint2 *>(lds_i32 + ((4 * (size_t)nb + 4 + 1) & ~(size_t)1));
    // stage_tables: the new tables are made HERE, in LDS behind the arrays above and in the generator's layout, and leave
    // in one coalesced piece - not as 18 scattered two-byte stores per node - and a chain whose next draw is small is
    // drawn by this workgroup itself, straight from them
    const bool stage = g.stage_tables != 0u;
    IdxT *const ltab = reinterpret_cast<IdxT *>(reinterpret_cast<char *>(step + nb) + ((16u - ((size_t)nb * 8u) % 16u) % 16u));
    IdxT *l_parent = ltab, *l_left = ltab + nb, *l_right = ltab + 2 * (size_t)nb, *l_nleaf = ltab + 3 * (size_t)nb,
         *l_depth = ltab + 4 * (size_t)nb, *l_tin = ltab + 5 * (size_t)nb, *l_first = ltab + 6 * (size_t)nb,
         *l_order = ltab + 7 * (size_t)nb, *l_up = ltab + 7 * (size_t)nb + g.leaf_order_len;
    const int32_t tid = (int32_t)threadIdx.x, nt = (int32_t)blockDim.x;
    auto phase = [&](uint32_t k) {
        if (g.prof && tid == 0 && j < 8u)
            g.prof[j * 8u + k] = wall_clock64();
    };
    phase(0);
    for (int32_t v = tid; v < nb; v += nt)
    {
        left[v] = (int32_t)t_left[v];
        right[v] = (int32_t)t_right[v];
        parent[v] = (int32_t)t_parent[v];
    }
    __syncthreads();
    phase(1); // tables loaded
    // the root: given with a re-root, otherwise the one node that is its own parent (device moves never re-root)
    if (is_ext)
    {
        if (tid == 0)
            shared[0] = g.ext[x].new_root;
    }
    else
        for (int32_t v = tid; v < nb; v += nt)
            if (parent[v] == v)
                shared[0] = v;
    const int32_t n_edits = is_ext ? g.ext[x].n_edits : g.info[cand].n_edits;
    const lvbgpu_edit_dev *ed = is_ext ? g.ext_edits + g.ext[x].edit_off : g.edits + (size_t)cand * g.stride_e;
    for (int32_t i = tid; i < n_edits; i += nt)
    {
        const lvbgpu_edit_dev e = ed[i];
        left[e.node] = e.left;
        right[e.node] = e.right;
    }
    __syncthreads();
    const int32_t root = shared[0];
    auto has_children = [&](int32_t v) { return v >= n || v == root; };
    for (int32_t v = tid; v < nb; v += nt)
    {
        if (has_children(v))
        {
            parent[left[v]] = v;
            parent[right[v]] = v;
        }
        nleaf[v] = has_children(v) ? 0 : 1;
    }
    __syncthreads();
    if (tid == 0)
        parent[root] = root; // nobody's child
    __syncthreads();
    // leaves below: every leaf counts itself into each of its ancestors (LDS atomics; no level-by-level rounds)
    for (int32_t v = tid; v < nb; v += nt)
        if (!has_children(v))
            for (int32_t y = v; y != root;)
            {
                y = parent[y];
                atomicAdd(&nleaf[y], 1);
            }
    __syncthreads();
    // What a node's root-ward walk adds at each step depends on the node stepped FROM only: from a right child the
    // preorder number skips the left sister's whole subtree (2 leaves-below, and as many leaf positions), from a left
    // child just the parent.  Packed once per node - {parent, weight}, weight = 2 nleaf[left sister] or 1 - the walk
    // below is one 8-byte LDS read per level instead of a chain of four dependent ones (parent, right[parent],
    // left[parent], nleaf[that]): the kernel's time was its depth (20-31 us on the start trees of a 500-taxon run,
    // 8 us on the shallow trees at its end) and it sits on the accept path of every annealing step.
    phase(2); // rewrites applied, parents, leaves below
    for (int32_t v = tid; v < nb; v += nt)
    {
        const int32_t p = parent[v];
        step[v] = make_uint2((uint32_t)p, (v != root && right[p] == v) ? 2u * (uint32_t)nleaf[left[p]] : 1u);
    }
    __syncthreads();
    // everything else is a function of the node's own path to the root: its length is the depth, the ancestors met at
    // distances 1, 2, 4, .. are the lifting table's entries (the root beyond), and the preorder number (left subtree
    // first) and the position of the first leaf below are sums along it (weight, and weight / 2: 1 / 2 = 0)
    if (stage && tid == 0)
        l_order[g.leaf_order_len - 1u] = (IdxT)0; // (one leaf is the root: the last place of the leaf order is never written)
    auto put = [&](IdxT *global_at, IdxT *lds_at, size_t at, IdxT value) __attribute__((always_inline)) {
        if (stage)
            lds_at[at] = value;
        else
            store_tab(global_at + at, value, through);
    };
    for (int32_t v = tid; v < nb; v += nt)
    {
        uint32_t tv = 0, fv = 0;
        int32_t d = 0, filled = 0;
        for (int32_t y = v; y != root;)
        {
            const uint2 st = step[y];
            tv += st.y;
            fv += st.y >> 1;
            y = (int32_t)st.x;
            d++;
            if ((d & (d - 1)) == 0 && filled < K)
                put(t_up, l_up, (size_t)filled++ * nb + v, (IdxT)y); // d = 2^filled
        }
        for (; filled < K; filled++)
            put(t_up, l_up, (size_t)filled * nb + v, (IdxT)root);
        put(t_parent, l_parent, (size_t)v, (IdxT)parent[v]);
        put(t_left, l_left, (size_t)v, (IdxT)(has_children(v) ? left[v] : 0));
        put(t_right, l_right, (size_t)v, (IdxT)(has_children(v) ? right[v] : 0));
        put(t_nleaf, l_nleaf, (size_t)v, (IdxT)nleaf[v]);
        put(t_depth, l_depth, (size_t)v, (IdxT)d);
        put(t_tin, l_tin, (size_t)v, (IdxT)tv);
        put(t_first, l_first, (size_t)v, (IdxT)fv);
        if (!has_children(v))
            put(t_order, l_order, (size_t)fv, (IdxT)v);
    }
    if (stage)
    {
        __syncthreads();
        phase(3); // root-ward walks done, tables in LDS
        // this chain's next draw, if it is a small one: drawn here and now, from the tables in LDS - no generator
        // workgroup has to wait for them to reach memory, fetch them again and start
        {
            int32_t fs = -1;
            for (uint32_t i = 0; i < gen.nseg; i++) // (nseg == 0: no generator in this launch)
                if (gen.seg[i].fused && gen.seg[i].chain == chain)
                    fs = (int32_t)i;
            if (fs >= 0)
            {
                const GenSeg &sg = gen.seg[fs];
                using P = typename TabPtr<IdxT, true>::type;
                const P lt = (P)ltab;
                Tab<IdxT, true> t;
                t.parent = lt;
                t.left = lt + nb;
                t.right = lt + 2 * (size_t)nb;
                t.nleaf = lt + 3 * (size_t)nb;
                t.depth = lt + 4 * (size_t)nb;
                t.tin = lt + 5 * (size_t)nb;
                t.first_leaf = lt + 6 * (size_t)nb;
                t.leaf_order = lt + 7 * (size_t)nb;
                t.up = lt + 7 * (size_t)nb + (size_t)g.leaf_order_len;
                t.n = n;
                t.nb = nb;
                t.root = sg.root;
                t.K = K;
                const uint32_t lane = threadIdx.x & 63u;
                const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
                for (uint32_t bl = wave; bl < sg.count; bl += nwaves)
                    generate_one(t, gen, sg, bl, lane);
            }
        }
        __syncthreads();
        phase(4); // drawn
        // the tables leave in one piece (eight bytes per store; written through where a generator of this launch waits)
        const unsigned long long *src8 = reinterpret_cast<const unsigned long long *>(ltab);
        unsigned long long *dst8 = reinterpret_cast<unsigned long long *>(tab);
        const uint32_t n8 = g.table_bytes / 8u;
        for (uint32_t i = (uint32_t)tid; i < n8; i += (uint32_t)nt)
            store_tab(dst8 + i, src8[i], through);
        phase(5); // copied out (issued)
    }
    if (through)
    {
        // every wave's stores acknowledged (written through: in no cache of this chip), then the chain's word
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0 && !g.withhold_ready)
            __hip_atomic_store(g.table_ready + chain, g.ready_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------------
// The POST launch: everything that lies between two scoring walks of an annealing step, in ONE launch of 1024-thread
// workgroups with three roles -
//   [0, n_reb)                the table rebuilds of the chains that moved (accepted device candidates: rewrites from the
//                             batch they were drawn in; re-roots: rewrites from the host), wave 0 of each first sending an
//                             accepted candidate's record to the host;
//   [n_reb, n_reb + n_cblk)   the commit walk of those moves (walk_body.hpp, 16 waves = 16 items per workgroup): accepted
//                             candidates' own device-built programs (picked by position) and host-built programs of a
//                             second block;
//   the rest                  the NEXT step's generator, a segment of which waits for its chain's rebuild (workgroups are
//                             dealt in order, so whoever is waited for is resident or done).
// Round 3 ran these as up to five launches on two streams with two event pairs (commit walk, rebuild + gather, re-root
// walk, its rebuild, generator): every kernel boundary costs 8-9 us whatever the kernel does, a cross-stream hand-over
// 2-5 us, and on the accept path the host's launch calls are what the device waits for.
static_assert(GEN_THREADS == REBUILD_THREADS, "one workgroup size for every role of the post launch");
// (A NARROW form - 4-wave workgroups with 38 KB of LDS, for a context whose post launch runs beside another context's
// scoring walk, where a 16-wave workgroup with 100+ KB finds no CU to run on - was built and measured in round 4: a post
// launch that started beside the other lane's walk took 81 us instead of 86, alone 53 instead of 36; the 32-chain run with
// it 1.225 s against 1.241 s without, three runs each.  Not kept: profiles/experiments/r04_post_launch_and_lanes.md.)
template <typename IdxT, bool WIDE>
__global__ __launch_bounds__(GEN_THREADS) void post_kernel(const PostArgs p)
{
    constexpr uint32_t WAVES = GEN_WAVES;
    extern __shared__ uint4 lds_dyn[];
    const uint32_t b = blockIdx.x;
    // LVBGPU_POST_PROFILE (tools/post_profile.py): when did each workgroup of the launch start and end, and in which role
    const unsigned long long t_start = p.prof ? wall_clock64() : 0ull;
    auto stamp = [&](uint32_t role) {
        if (p.prof && threadIdx.x == 0 && b < 1000u)
        {
            p.prof[1u + 4u * b] = role;
            p.prof[2u + 4u * b] = t_start;
            p.prof[4u + 4u * b] = wall_clock64();
            if (b == 0)
                p.prof[0] = gridDim.x;
        }
    };
    if (b < p.n_reb)
    {
        rebuild_role<IdxT>(p.reb, p.gat, p.gen, b, reinterpret_cast<int32_t *>(lds_dyn));
        __syncthreads();
        stamp(1);
        return;
    }
    if (b < p.n_reb + p.n_cblk)
    {
        // Which item: the commit walk of tile group g runs on the XCD whose L2 holds that tile's column slice - the one
        // the scoring walks put it on (fitch_walk: XCD x takes the x-th eighth of the tile-major item list, so group g
        // lies with XCD floor((2 g + 1) 4 / ngroups)).  The accepted candidate's rows were read there a moment ago, and
        // the sets written here are read there by the next scoring walk; dealt without regard to it, a commit wave's
        // every load went to memory (one chain: 21 us for 25 tokens) and so did the next walk's reads of the new sets.
        // The hardware deals consecutive workgroups round-robin over the XCDs; n_cblk is a multiple of 8.
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const uint32_t cb = b - p.n_reb, x = b & 7u;
        const uint32_t rank = (cb - ((x - (p.n_reb & 7u)) & 7u)) >> 3; // this workgroup's place among the commit role's on XCD x
        const uint32_t ng = p.commit.ngroups, k = p.commit.B;
        const uint32_t g_lo = (x * ng + 3u) >> 3, g_hi = x == 7u ? ng : ((x + 1u) * ng + 3u) >> 3;
        const uint32_t q = rank * WAVES + wave;
        if (q < (g_hi - g_lo) * k)
        {
            const uint32_t g = g_lo + q / k, c = q - (q / k) * k;
            walk_item<true, WIDE, 0>(p.commit, lds_dyn, lane, wave, WAVES, g * k + c);
        }
        if (p.prof)
        {
            __syncthreads();
            stamp(2);
        }
        return;
    }
    if (p.gen.nseg)
    {
        const uint32_t gb = b - p.n_reb - p.n_cblk;
        gen_role<IdxT, true>(p.gen, gb, p.gen.n_gen_blocks, lds_dyn);
        if (p.prof)
        {
            __syncthreads();
            stamp(3);
        }
    }
}

// the dynamic-LDS ceiling is an attribute of the function ON A DEVICE: raised once for each device a context of this
// process launches on (the walk does the same per context, raise_lds_limit), whichever host thread gets there first
static hipError_t raise_generator_lds()
{
    static hipError_t raised_on[64];
    static std::once_flag asked_on[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64)
        return hipErrorInvalidDevice;
    std::call_once(asked_on[dev], [dev] {
        hipError_t e = hipSuccess;
        for (const void *f : {reinterpret_cast<const void *>(&propose_kernel<uint16_t, true>),
                              reinterpret_cast<const void *>(&propose_kernel<int32_t, true>),
                              reinterpret_cast<const void *>(&post_kernel<uint16_t, false>),
                              reinterpret_cast<const void *>(&post_kernel<uint16_t, true>),
                              reinterpret_cast<const void *>(&post_kernel<int32_t, false>),
                              reinterpret_cast<const void *>(&post_kernel<int32_t, true>)})
            if (e == hipSuccess)
                e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS_BYTES);
        raised_on[dev] = e;
    });
    return raised_on[dev];
}

static size_t rebuild_lds_bytes(int32_t nb)
{
    return (((size_t)4 * nb + 4 + 1) & ~(size_t)1) * sizeof(int32_t) + (size_t)nb * sizeof(uint2);
}

// one candidate per wave while the chip has room; beyond that waves take several.  Every segment gets workgroups in
// proportion to its candidates (at least one).  per_cu: generator workgroups a CU can hold.  Returns their number.
static uint32_t deal_generator_blocks(GenArgs &g, uint32_t per_cu, uint32_t waves = GEN_WAVES)
{
    uint32_t total = 0;
    for (uint32_t s = 0; s < g.nseg; s++)
        total += g.seg[s].count;
    const uint32_t budget = 256u * per_cu;
    const uint32_t want_all = (total + waves - 1) / waves;
    uint32_t nblk = 0;
    for (uint32_t s = 0; s < g.nseg; s++)
    {
        const uint32_t want = (g.seg[s].count + waves - 1) / waves;
        uint32_t give = want_all <= budget ? want : (uint32_t)((uint64_t)want * budget / want_all);
        give = std::max(1u, std::min(give, std::max(want, 1u)));
        if (g.seg[s].fused)
            give = 0; // (drawn by its chain's rebuilding workgroup)
        g.seg[s].blk_start = nblk;
        nblk += give;
    }
    return nblk;
}

hipError_t launch_propose(const GenArgs &args, hipStream_t stream)
{
    if (args.nseg == 0 || args.nseg > MAX_GEN_SEGS)
        return hipErrorInvalidValue;
    const size_t lds_max = raise_generator_lds() == hipSuccess ? (size_t)MAX_LDS_BYTES : (size_t)64 * 1024;
    GenArgs g = args;
    uint32_t total = 0;
    for (uint32_t s = 0; s < g.nseg; s++)
    {
        total += g.seg[s].count;
        g.seg[s].wait = 0;
        g.seg[s].fused = 0;
    }
    if (total == 0)
        return hipSuccess;
    const size_t pair_lds = g.pairs ? PAIR_LDS_BYTES + 16 : 0; // (behind the tables, 16-byte aligned)
    g.use_lds = g.table_bytes + pair_lds <= lds_max ? 1 : 0; // else the tables are read where they lie (L2-resident)
    size_t lds = (g.use_lds ? g.table_bytes : 0) + pair_lds;
    const uint32_t per_cu = g.use_lds ? (uint32_t)std::max<size_t>(1, std::min<size_t>(2, lds_max / std::max<size_t>(lds, 1))) : 2u;
    const uint32_t nblk = g.n_gen_blocks = deal_generator_blocks(g, per_cu);
    const dim3 grid(nblk), block(GEN_THREADS);
    if (g.idx_bytes == 2)
    {
        if (g.use_lds)
            hipLaunchKernelGGL((propose_kernel<uint16_t, true>), grid, block, lds, stream, g);
        else
            hipLaunchKernelGGL((propose_kernel<uint16_t, false>), grid, block, lds, stream, g);
    }
    else if (g.use_lds)
        hipLaunchKernelGGL((propose_kernel<int32_t, true>), grid, block, lds, stream, g);
    else
        hipLaunchKernelGGL((propose_kernel<int32_t, false>), grid, block, lds, stream, g);
    return hipGetLastError();
}

bool post_can_generate(const GenArgs &g)
{
    return g.nseg >= 1 && g.nseg <= MAX_GEN_SEGS && g.moves == nullptr && raise_generator_lds() == hipSuccess &&
           g.table_bytes + (g.pairs ? PAIR_LDS_BYTES + 16 : 0) <= MAX_LDS_BYTES;
}

hipError_t launch_post(const PostArgs &args, hipStream_t stream)
{
    PostArgs p = args;
    if (raise_generator_lds() != hipSuccess)
        return hipErrorInvalidValue;
    const uint32_t waves = GEN_WAVES;
    size_t lds = 0;
    p.reb.stage_tables = 0;
    if (p.n_reb)
    {
        lds = rebuild_lds_bytes(p.reb.nb);
        if (lds > MAX_LDS_BYTES || p.reb.n_pick > p.n_reb)
            return hipErrorInvalidValue;
        // the new tables staged in LDS behind the rebuild's arrays (16-byte aligned), where they fit
        const uint32_t tb = p.gen.nseg ? p.gen.table_bytes : p.reb.table_bytes;
        if (tb && lds + 16 + tb <= MAX_LDS_BYTES && (p.reb.rebuild_picks || p.n_reb > p.reb.n_pick))
        {
            p.reb.stage_tables = 1;
            p.reb.table_bytes = tb;
            lds += 16 + tb;
        }
    }
    // a chain that is rebuilt here and draws few candidates next is drawn by its rebuilding workgroup (not with pairing:
    // pairs are made by generating workgroups)
    for (uint32_t s = 0; s < p.gen.nseg; s++)
    {
        p.gen.seg[s].fused = 0;
        if (p.reb.stage_tables && p.gen.seg[s].wait && !p.gen.pairs && !p.gen.prof && p.gen.seg[s].count <= 4u * waves)
        {
            p.gen.seg[s].fused = 1;
            p.gen.seg[s].wait = 0;
        }
    }
    p.reb.wait_mask = 0;
    for (uint32_t s = 0; s < p.gen.nseg; s++)
        if (p.gen.seg[s].wait)
            p.reb.wait_mask |= 1ull << p.gen.seg[s].chain;
    uint32_t gen_total = 0, gen_blocks = 0;
    if (p.gen.nseg)
    {
        if (!post_can_generate(p.gen))
            return hipErrorInvalidValue;
        for (uint32_t s = 0; s < p.gen.nseg; s++)
            gen_total += p.gen.seg[s].count;
        if (gen_total == 0)
            p.gen.nseg = 0;
    }
    if (p.gen.nseg)
    {
        p.gen.use_lds = 1;
        lds = std::max(lds, (size_t)p.gen.table_bytes + (p.gen.pairs ? PAIR_LDS_BYTES + 16 : 0));
    }
    // the commit walk's parked sets take what LDS is left - all of it while the generator's workgroups are few (one per
    // CU is plenty then), half of it when they are many (cold chains drawing a thousand candidates each: two generator
    // workgroups per CU matter more than the commit walk's bursts)
    p.n_cblk = 0;
    if (p.commit.nitems)
    {
        const bool many = (gen_total + waves - 1) / waves > 256u;
        const size_t budget = many ? MAX_LDS_BYTES / 2 : MAX_LDS_BYTES;
        size_t clds = 0;
        hipError_t e = shape_walk(p.commit, true, waves, budget, &clds);
        if (e != hipSuccess && many)
            e = shape_walk(p.commit, true, waves, MAX_LDS_BYTES, &clds);
        if (e != hipSuccess)
            return e;
        p.commit.flip = 0;
        p.commit.watcher = 0;
        lds = std::max(lds, clds);
        // (every XCD walks the tile groups whose column slices its L2 holds: as many workgroups per XCD as the fullest needs)
        {
            const uint32_t ng = p.commit.ngroups, k = p.commit.B;
            uint32_t most = 0;
            for (uint32_t x = 0; x < 8u; x++)
            {
                const uint32_t g_lo = (x * ng + 3u) >> 3, g_hi = x == 7u ? ng : ((x + 1u) * ng + 3u) >> 3;
                most = std::max(most, (g_hi - g_lo) * k);
            }
            p.n_cblk = 8u * ((most + waves - 1) / waves);
        }
    }
    if (p.gen.nseg)
    {
        gen_blocks = p.gen.n_gen_blocks =
            deal_generator_blocks(p.gen, (uint32_t)std::max<size_t>(1, std::min<size_t>(2, MAX_LDS_BYTES / std::max<size_t>(lds, 1))), waves);
    }
    else
        p.gen.n_gen_blocks = 0;
    const uint32_t nblk = p.n_reb + p.n_cblk + gen_blocks;
    if (nblk == 0)
        return hipSuccess;
    const bool wide = p.commit.nitems != 0 && walk_needs_wide(p.commit);
    const dim3 grid(nblk), block(waves * 64u);
    const bool idx16 = (p.gen.nseg ? p.gen.idx_bytes : p.reb.idx_bytes) != 4;
    if (idx16)
    {
        if (wide)
            hipLaunchKernelGGL((post_kernel<uint16_t, true>), grid, block, lds, stream, p);
        else
            hipLaunchKernelGGL((post_kernel<uint16_t, false>), grid, block, lds, stream, p);
    }
    else if (wide)
        hipLaunchKernelGGL((post_kernel<int32_t, true>), grid, block, lds, stream, p);
    else
        hipLaunchKernelGGL((post_kernel<int32_t, false>), grid, block, lds, stream, p);
    return hipGetLastError();
}

} // namespace lvbgpu
