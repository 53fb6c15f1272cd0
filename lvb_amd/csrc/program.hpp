// program.hpp - host side of the scoring path: turn "a tree + which nodes are dirty" into the
// token program one wavefront walks (lvb_amd/csrc/fitch_kernels.hip).
//
// What it replaces in the reference: getplen's own scheduling of dirty nodes
// (TreeEvaluation.c:191-236: index-order todo list, swept repeatedly until every dirty node
// has had both children ready) and the dirty marking done by the proposal generators
// (TreeOperations.c:88-103 make_dirty_below and its call sites 207, 302, 330-334, 413, 520-537,
// 631-635).  Here the dirty set is derived from the edits (edited nodes + their ancestors below
// the root) and emitted once, in true postorder, so the device never sweeps.
//
// Token program (uint32 per token, every token names exactly one state-set row to load):
//   bits  0..23  row index (0..n-1 leaf rows, n..2n-4 resident node rows, or a staging index)
//   bits 24..29  number of merges after this token's own step: acc = combine(pop(), acc)
//   bit  30      FRESH: this token starts a chain, acc = row (no combine)
//   bit  31      PUSH : save acc on the operand stack before starting the chain (implies FRESH)
//   a token without FRESH means acc = combine(acc, row)
// Every combine (non-FRESH token or merge) produces one node, in the order of dsts[]; the last
// two combines of a program are the root's (children, then the root leaf's own row) and have
// dst -1 (TreeEvaluation.c:238-264: counted, never stored).
// A program for D dirty nodes has D + 3 tokens and D + 2 combines.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace lvbgpu
{

constexpr uint32_t TOK_ROW_MASK = 0x00FFFFFFu;
constexpr uint32_t TOK_MERGE_SHIFT = 24;
constexpr uint32_t TOK_MERGE_MASK = 0x3Fu;
constexpr uint32_t TOK_FRESH = 1u << 30;
constexpr uint32_t TOK_PUSH = 1u << 31;
constexpr int32_t UNSET = -1;
constexpr int MAX_ROWS = 1 << 24;

struct Edit
{
    int32_t node, left, right;
};

// child/parent arrays of an unrooted binary tree rooted at leaf `root` (reference layout:
// LVB.h:121-128 scalars, without the state sets)
struct Topology
{
    int32_t n = 0;  // taxa
    int32_t nb = 0; // 2n-3 records
    int32_t root = 0;
    std::vector<int32_t> parent, left, right;

    // set from child arrays; parents are derived.  Returns false (with why) unless the arrays
    // describe one binary tree over all 2n-3 records rooted at leaf `root`.
    bool assign(int32_t n_taxa, const int32_t *l, const int32_t *r, int32_t root_leaf, std::string *why);
    bool validate(std::string *why) const;
};

struct Program
{
    std::vector<uint32_t> toks;
    std::vector<int32_t> dsts; // one per combine
    int32_t dirty = 0;         // D
    int32_t max_stack = 0;
};

// Reusable scratch for building many programs against one topology.
class ProgramBuilder
{
  public:
    explicit ProgramBuilder(int32_t nb = 0) { resize(nb); }
    void resize(int32_t nb);

    // Incremental candidate: apply `edits` (+ optional new root) to `topo` temporarily, derive
    // the dirty set, append the program to `out`.  `topo` is restored before returning.
    // Returns false with *why set when the edits do not give a tree.
    bool build_candidate(Topology &topo, const Edit *edits, int32_t n_edits, int32_t new_root, Program &out,
                         std::string *why);

    // Apply edits permanently (commit).  The dirty list of the edit is left in dirty_list().
    bool apply_edits(Topology &topo, const Edit *edits, int32_t n_edits, int32_t new_root, std::string *why);

    // Every internal node dirty (full evaluation).
    void build_full(const Topology &topo, Program &out);

    // Arbitrary dirty flags (strict compat: the reference's sitestate[0]==0 convention); dirty
    // nodes whose parent is clean are evaluated and stored but feed nothing.
    void build_flagged(const Topology &topo, const uint8_t *dirty, Program &out);

    const std::vector<int32_t> &dirty_list() const { return dirty_list_; }

  private:
    struct Undo
    {
        int32_t kind, idx, old;
    };
    struct Frame
    {
        int32_t v, stage, first, second;
    };
    bool is_dirty(int32_t v) const { return mark_[v] == epoch_; }
    void next_epoch();
    bool apply(Topology &topo, const Edit *edits, int32_t n_edits, int32_t new_root, std::string *why);
    void undo(Topology &topo);
    bool mark_from_edits(const Topology &topo, const Edit *edits, int32_t n_edits, std::string *why);
    void compute_need(const Topology &topo, const std::vector<int32_t> &tops);
    void emit_subtree(const Topology &topo, int32_t top, Program &out);
    void emit_rooted(const Topology &topo, Program &out);
    void tok_row(Program &out, int32_t row, bool fresh);
    void tok_merge(Program &out, int32_t dst);

    std::vector<uint32_t> mark_;
    std::vector<int32_t> need_;
    std::vector<int32_t> dirty_list_;
    std::vector<int32_t> order_;
    std::vector<Undo> undo_;
    std::vector<Frame> frames_;
    uint32_t epoch_ = 0;
    bool acc_live_ = false;
    int32_t depth_ = 0;
};

} // namespace lvbgpu
