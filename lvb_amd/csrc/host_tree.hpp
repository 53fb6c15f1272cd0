// host_tree.hpp - the object behind lvbhost_tree (include/lvbhost.h)
#pragma once

#include <cstdint>
#include <unordered_map>
#include <vector>

#include "program.hpp"
#include "proposals.hpp"

// The distinct best topologies found so far: what the reference keeps in its treestack
// (Treestack.c:231-306 CompareTreeToTreestack: push only if the topology is new; it compares the trees' sorted
// object sets exactly, TreeOperations.c:1392-1498).  Here a topology is looked up by a hash of its bipartitions
// (XOR of per-taxon keys of one side, the numerically smaller of the two complementary keys: independent of
// rooting and node numbering) and, on a hash hit, compared EXACTLY through its canonical form: the tree
// re-rooted at taxon 0 with every node's subtrees ordered by their smallest taxon, written out in preorder
// (-1 opens an internal node).  Two trees have the same canonical form iff they are the same unrooted topology,
// so the count cannot differ from the reference's by a collision.  Every distinct tree is kept (the reference's
// treestack is unbounded too).
struct BestSet
{
    std::vector<uint64_t> key;            // per taxon
    struct Kept
    {
        std::vector<int32_t> left, right;
        int32_t root;
        std::vector<int32_t> canon; // empty until a second tree has to be told apart from this one (lazy)
    };
    std::vector<Kept> kept;                                   // every distinct topology, in order of arrival
    std::unordered_multimap<uint64_t, size_t> by_hash;        // hash -> index into kept

    void reset(int32_t n);
    void clear()
    {
        by_hash.clear();
        kept.clear();
    }
    size_t count() const { return kept.size(); }
    uint64_t hash(const lvbgpu::Topology &t, std::vector<uint64_t> &scratch) const;
    static void canonical(const lvbgpu::Topology &t, std::vector<int32_t> &out);
    bool insert(const lvbgpu::Topology &t); // true if the topology is new
    void index_all();                       // canonical forms and hashes of the trees kept without them
    Kept pop_last();                        // take the newest tree off (and out of the index)
    void push_kept(Kept &&k, std::vector<uint64_t> &scratch); // put one back without comparing
};

struct lvbhost_tree
{
    lvbgpu::Topology topo;
    lvbgpu::Rng rng;
    lvbgpu::ProgramBuilder pb;
    std::vector<lvbgpu::Edit> scratch;
    BestSet best;
};
