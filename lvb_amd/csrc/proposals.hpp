// proposals.hpp - host-side neighbourhood generators that emit candidates as EDITS against a
// topology instead of copying the tree (the reference's mutate_* start with a full treecopy,
// TreeOperations.c:171/253/364 -> 737-773, 25 MB per proposal at 500 x 50k).
//
// Semantics follow the reference's generators so the dirty sets are the same:
//   propose_nni  <- mutate_nni  TreeOperations.c:160-209
//   propose_spr  <- mutate_spr  TreeOperations.c:236-335 (the pruned parent is re-used as the graft node)
//   propose_tbr  <- mutate_tbr  TreeOperations.c:337-541 (SPR + re-root of the moved subtree at a leaf edge)
//   reroot_edits <- lvb_reroot  TreeOperations.c:576-637 (as edits along the old-root..new-root path only)
//   random_topology <- PullRandomTree/GenerateRandomTopology TreeOperations.c:799-811, 957-1011
// The random stream is our own (xorshift64*), not the reference's Marsaglia UNI.
#pragma once

#include <cstdint>
#include <vector>

#include "program.hpp"

namespace lvbgpu
{

struct Rng
{
    uint64_t s;
    explicit Rng(uint64_t seed = 0x9E3779B97F4A7C15ull) : s(seed ? seed : 0x9E3779B97F4A7C15ull) {}
    uint64_t next()
    {
        s ^= s >> 12;
        s ^= s << 25;
        s ^= s >> 27;
        return s * 0x2545F4914F6CDD1Dull;
    }
    // uniform integer in [0, n)
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

enum MoveKind
{
    MOVE_NNI = 0,
    MOVE_SPR = 1,
    MOVE_TBR = 2
};

// One random move as its parameters (the layout of lvbgpu_move in include/lvbgpu.h: NNI a = u, b = 1 to
// give away u's right child; SPR a = src, b = dest; TBR a = src, b = dest, c = leaf to re-root the
// moved subtree at, -1 = as SPR), and parameters -> rewrites.  A search can hand the parameters to the
// device (lvbgpu_score_moves) and only turn the accepted move into rewrites.
struct MoveParams
{
    int32_t kind, a, b, c;
};
MoveParams draw_move(const Topology &t, int kind, Rng &rng);
int move_edits(const Topology &t, const MoveParams &m, std::vector<Edit> &out);

// Each appends the child-pair rewrites of ONE random move to `out` and returns how many.
int propose_nni(const Topology &t, Rng &rng, std::vector<Edit> &out);
int propose_spr(const Topology &t, Rng &rng, std::vector<Edit> &out);
int propose_tbr(const Topology &t, Rng &rng, std::vector<Edit> &out);
int propose(const Topology &t, int kind, Rng &rng, std::vector<Edit> &out);

// deterministic forms (tests drive these with the reference's own choices)
int nni_edits(const Topology &t, int32_t u, bool swap_right, std::vector<Edit> &out);
int spr_edits(const Topology &t, int32_t src, int32_t dest, std::vector<Edit> &out);
int tbr_edits(const Topology &t, int32_t src, int32_t dest, int32_t newroot_leaf, std::vector<Edit> &out);
bool spr_move_allowed(const Topology &t, int32_t src, int32_t dest);
// leaves under `top`, left before right (the order of the reference's addtoarray, TreeOperations.c:560-574)
void subtree_leaves(const Topology &t, int32_t top, std::vector<int32_t> &leaves);

// edits that re-root the tree at leaf `newroot` (old root leaf ends with children (-1,-1))
int reroot_edits(const Topology &t, int32_t newroot, std::vector<Edit> &out);

// random start topology: leaves sprout (Yule shape), taxa assigned to leaves at random, root = leaf 0
void random_topology(int32_t n, Rng &rng, Topology &out);

} // namespace lvbgpu
