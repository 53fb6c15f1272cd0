// host_tree.hpp - the object behind lvbhost_tree (include/lvbhost.h)
#pragma once

#include <cstdint>
#include <unordered_set>
#include <vector>

#include "program.hpp"
#include "proposals.hpp"

// The distinct best topologies found so far: what the reference keeps in its treestack
// (Treestack.c:231-306 CompareTreeToTreestack: push only if the topology is new).  A topology is
// identified by the set of its bipartitions, each hashed from the XOR of per-taxon keys of one
// side (the numerically smaller of the two complementary keys), so the identity does not depend on
// rooting or node numbering.
struct BestSet
{
    std::vector<uint64_t> key;            // per taxon
    std::unordered_set<uint64_t> seen;    // topology hashes
    struct Kept
    {
        std::vector<int32_t> left, right;
        int32_t root;
    };
    std::vector<Kept> kept;               // the first `cap` of them, for output
    size_t cap = 1024;

    void reset(int32_t n);
    void clear()
    {
        seen.clear();
        kept.clear();
    }
    uint64_t hash(const lvbgpu::Topology &t, std::vector<uint64_t> &scratch) const;
    bool insert(const lvbgpu::Topology &t); // true if the topology is new
};

struct lvbhost_tree
{
    lvbgpu::Topology topo;
    lvbgpu::Rng rng;
    lvbgpu::ProgramBuilder pb;
    std::vector<lvbgpu::Edit> scratch;
    BestSet best;
};
