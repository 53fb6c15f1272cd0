// program.cpp - see program.hpp.  Pure host C++ (no HIP): also built into liblvbhost.so so the
// host logic is testable on a machine without a GPU.
#include "program.hpp"

#include <algorithm>

namespace lvbgpu
{

// ------------------------------------------------------------------------------ Topology

bool Topology::assign(int32_t n_taxa, const int32_t *l, const int32_t *r, int32_t root_leaf, std::string *why)
{
    n = n_taxa;
    nb = 2 * n_taxa - 3;
    root = root_leaf;
    left.assign(l, l + nb);
    right.assign(r, r + nb);
    parent.assign(nb, UNSET);
    for (int32_t v = 0; v < nb; v++)
    {
        const int32_t a = left[v], b = right[v];
        if ((a < 0) != (b < 0))
        {
            if (why)
                *why = "node " + std::to_string(v) + " has exactly one child";
            return false;
        }
        if (a < 0)
            continue;
        if (a >= nb || b >= nb || a == b || a == v || b == v)
        {
            if (why)
                *why = "node " + std::to_string(v) + " has out-of-range or repeated children";
            return false;
        }
        if (parent[a] != UNSET || parent[b] != UNSET)
        {
            if (why)
                *why = "a child of node " + std::to_string(v) + " already has a parent";
            return false;
        }
        parent[a] = v;
        parent[b] = v;
    }
    return validate(why);
}

bool Topology::validate(std::string *why) const
{
    auto fail = [&](const std::string &s) {
        if (why)
            *why = s;
        return false;
    };
    if (n < 3 || nb != 2 * n - 3)
        return fail("need at least 3 taxa");
    if (root < 0 || root >= n)
        return fail("root must be a leaf index");
    if (parent[root] != UNSET)
        return fail("root has a parent");
    for (int32_t v = 0; v < nb; v++)
    {
        const bool has_children = left[v] >= 0;
        if (v < n && v != root && has_children)
            return fail("leaf " + std::to_string(v) + " has children but is not the root");
        if ((v >= n || v == root) && !has_children)
            return fail("node " + std::to_string(v) + " lacks children");
        if (v != root && parent[v] == UNSET)
            return fail("node " + std::to_string(v) + " is not in the tree");
    }
    // reachability (rules out a cycle detached from the root)
    std::vector<int32_t> stack{root};
    int32_t seen = 0;
    while (!stack.empty())
    {
        const int32_t v = stack.back();
        stack.pop_back();
        if (++seen > nb)
            return fail("cycle in tree");
        if (left[v] >= 0)
        {
            stack.push_back(left[v]);
            stack.push_back(right[v]);
        }
    }
    if (seen != nb)
        return fail("tree does not span all records");
    return true;
}

// ------------------------------------------------------------------------------ builder

void ProgramBuilder::resize(int32_t nb)
{
    mark_.assign(nb, 0);
    need_.assign(nb, 0);
    epoch_ = 0;
}

void ProgramBuilder::next_epoch()
{
    if (++epoch_ == 0)
    {
        std::fill(mark_.begin(), mark_.end(), 0u);
        epoch_ = 1;
    }
    dirty_list_.clear();
}

enum
{
    U_LEFT,
    U_RIGHT,
    U_PARENT,
    U_ROOT
};

bool ProgramBuilder::apply(Topology &t, const Edit *edits, int32_t n_edits, int32_t new_root, std::string *why)
{
    undo_.clear();
    auto fail = [&](const std::string &s) {
        if (why)
            *why = s;
        return false;
    };
    for (int32_t i = 0; i < n_edits; i++)
    {
        const Edit &e = edits[i];
        if (e.node < 0 || e.node >= t.nb)
            return fail("edit names node out of range");
        if ((e.left < 0) != (e.right < 0))
            return fail("edit gives exactly one child");
        if (e.left >= 0 && (e.left >= t.nb || e.right >= t.nb || e.left == e.right || e.left == e.node ||
                            e.right == e.node))
            return fail("edit has out-of-range or repeated children");
        undo_.push_back({U_LEFT, e.node, t.left[e.node]});
        undo_.push_back({U_RIGHT, e.node, t.right[e.node]});
        t.left[e.node] = e.left;
        t.right[e.node] = e.right;
        if (e.left >= 0)
        {
            undo_.push_back({U_PARENT, e.left, t.parent[e.left]});
            t.parent[e.left] = e.node;
            undo_.push_back({U_PARENT, e.right, t.parent[e.right]});
            t.parent[e.right] = e.node;
        }
    }
    if (new_root >= 0 && new_root != t.root)
    {
        if (new_root >= t.n)
            return fail("new root is not a leaf");
        undo_.push_back({U_ROOT, 0, t.root});
        t.root = new_root;
        undo_.push_back({U_PARENT, new_root, t.parent[new_root]});
        t.parent[new_root] = UNSET;
    }
    // Local consistency, O(#edits): the edits must MOVE children, not duplicate or drop them.
    //  - a child that got a new parent must have been released by its old one;
    //  - a child released by an edited node must have been adopted by another edit.
    for (const Undo &u : undo_)
    {
        if (u.kind == U_PARENT)
        {
            const int32_t c = u.idx, was = u.old, now = t.parent[c];
            if (was >= 0 && was != now && (t.left[was] == c || t.right[was] == c))
                return fail("node " + std::to_string(c) + " would have two parents");
        }
        else if (u.kind == U_LEFT || u.kind == U_RIGHT)
        {
            const int32_t v = u.idx, oc = u.old;
            if (oc >= 0 && t.left[v] != oc && t.right[v] != oc && t.parent[oc] == v)
                return fail("node " + std::to_string(oc) + " would be left without a parent");
        }
    }
    return true;
}

void ProgramBuilder::undo(Topology &t)
{
    for (auto it = undo_.rbegin(); it != undo_.rend(); ++it)
    {
        switch (it->kind)
        {
        case U_LEFT: t.left[it->idx] = it->old; break;
        case U_RIGHT: t.right[it->idx] = it->old; break;
        case U_PARENT: t.parent[it->idx] = it->old; break;
        default: t.root = it->old; break;
        }
    }
    undo_.clear();
}

bool ProgramBuilder::mark_from_edits(const Topology &t, const Edit *edits, int32_t n_edits, std::string *why)
{
    auto fail = [&](const std::string &s) {
        if (why)
            *why = s;
        return false;
    };
    if (t.left[t.root] < 0)
        return fail("root has no children after the edits");
    for (int32_t i = 0; i < n_edits; i++)
    {
        int32_t v = edits[i].node;
        if (v < t.n)
        {
            // a leaf record: either the root (children given) or an ordinary leaf (none)
            if ((v == t.root) != (edits[i].left >= 0))
                return fail("leaf edit inconsistent with the root");
            continue;
        }
        if (edits[i].left < 0)
            return fail("internal node edited to have no children");
        // the edited node and its ancestors below the root (make_dirty_below, TreeOperations.c:88-103)
        int32_t steps = 0;
        while (v != t.root)
        {
            if (v < 0)
                return fail("edited node is not connected to the root");
            if (v < t.n)
                return fail("a leaf other than the root has children");
            if (mark_[v] == epoch_)
                break;
            mark_[v] = epoch_;
            dirty_list_.push_back(v);
            v = t.parent[v];
            if (++steps > t.nb)
                return fail("cycle among edited nodes");
        }
    }
    return true;
}

void ProgramBuilder::compute_need(const Topology &t, const std::vector<int32_t> &tops)
{
    // preorder over the dirty forest, then children-before-parents: Sethi-Ullman numbers, so the
    // child needing the deeper operand stack is evaluated first
    order_.clear();
    std::vector<int32_t> &st = order_; // reuse as output; separate stack below
    std::vector<int32_t> work(tops.begin(), tops.end());
    while (!work.empty())
    {
        const int32_t v = work.back();
        work.pop_back();
        st.push_back(v);
        const int32_t a = t.left[v], b = t.right[v];
        if (a >= t.n && is_dirty(a))
            work.push_back(a);
        if (b >= t.n && is_dirty(b))
            work.push_back(b);
    }
    for (auto it = order_.rbegin(); it != order_.rend(); ++it)
    {
        const int32_t v = *it;
        const int32_t a = t.left[v], b = t.right[v];
        const bool da = a >= t.n && is_dirty(a), db = b >= t.n && is_dirty(b);
        if (da && db)
        {
            const int32_t hi = std::max(need_[a], need_[b]), lo = std::min(need_[a], need_[b]);
            need_[v] = std::max(hi, lo + 1);
        }
        else if (da)
            need_[v] = need_[a];
        else if (db)
            need_[v] = need_[b];
        else
            need_[v] = 0;
    }
}

void ProgramBuilder::tok_row(Program &out, int32_t row, bool fresh)
{
    uint32_t tk = (uint32_t)row;
    if (fresh)
    {
        tk |= TOK_FRESH;
        if (acc_live_)
        {
            tk |= TOK_PUSH;
            depth_++;
            out.max_stack = std::max(out.max_stack, depth_);
        }
        acc_live_ = true;
    }
    out.toks.push_back(tk);
}

void ProgramBuilder::tok_merge(Program &out, int32_t dst)
{
    out.toks.back() += 1u << TOK_MERGE_SHIFT;
    depth_--;
    out.dsts.push_back(dst);
}

// evaluate the dirty subtree hanging under `top` (top itself dirty, or top == root: the
// "virtual" node above the root's two children whose result is never stored)
void ProgramBuilder::emit_subtree(const Topology &t, int32_t top, Program &out)
{
    frames_.clear();
    frames_.push_back({top, 0, 0, 0});
    while (!frames_.empty())
    {
        Frame &f = frames_.back();
        const int32_t v = f.v;
        const int32_t dst = (v == t.root) ? -1 : v;
        if (f.stage == 0)
        {
            const int32_t a = t.left[v], b = t.right[v];
            const bool da = a >= t.n && is_dirty(a), db = b >= t.n && is_dirty(b);
            if (!da && !db)
            {
                tok_row(out, a, true);
                tok_row(out, b, false);
                out.dsts.push_back(dst);
                frames_.pop_back();
            }
            else if (da && db)
            {
                const bool a_first = need_[a] >= need_[b];
                f.first = a_first ? a : b;
                f.second = a_first ? b : a;
                f.stage = 2;
                frames_.push_back({f.first, 0, 0, 0});
            }
            else
            {
                f.first = da ? a : b;
                f.second = da ? b : a; // the clean one
                f.stage = 1;
                frames_.push_back({f.first, 0, 0, 0});
            }
        }
        else if (f.stage == 1)
        {
            tok_row(out, f.second, false);
            out.dsts.push_back(dst);
            frames_.pop_back();
        }
        else if (f.stage == 2)
        {
            f.stage = 3;
            const int32_t second = f.second;
            frames_.push_back({second, 0, 0, 0});
        }
        else
        {
            tok_merge(out, dst);
            frames_.pop_back();
        }
    }
}

void ProgramBuilder::emit_rooted(const Topology &t, Program &out)
{
    emit_subtree(t, t.root, out);
    // finally the root leaf's own row (TreeEvaluation.c:255-263)
    tok_row(out, t.root, false);
    out.dsts.push_back(-1);
}

bool ProgramBuilder::build_candidate(Topology &t, const Edit *edits, int32_t n_edits, int32_t new_root,
                                     Program &out, std::string *why)
{
    if ((int32_t)mark_.size() != t.nb)
        resize(t.nb);
    next_epoch();
    bool ok = apply(t, edits, n_edits, new_root, why);
    if (ok)
        ok = mark_from_edits(t, edits, n_edits, why);
    if (ok)
    {
        std::vector<int32_t> tops;
        const int32_t a = t.left[t.root], b = t.right[t.root];
        if (a >= t.n && is_dirty(a))
            tops.push_back(a);
        if (b >= t.n && is_dirty(b))
            tops.push_back(b);
        compute_need(t, tops);
        const size_t tok0 = out.toks.size(), dst0 = out.dsts.size();
        acc_live_ = false;
        depth_ = 0;
        emit_rooted(t, out);
        out.dirty = (int32_t)dirty_list_.size();
        if (out.dsts.size() - dst0 != dirty_list_.size() + 2 || out.toks.size() - tok0 != dirty_list_.size() + 3)
        {
            if (why)
                *why = "edits leave dirty nodes unreachable from the root";
            ok = false;
        }
    }
    undo(t);
    return ok;
}

bool ProgramBuilder::apply_edits(Topology &t, const Edit *edits, int32_t n_edits, int32_t new_root, std::string *why)
{
    if ((int32_t)mark_.size() != t.nb)
        resize(t.nb);
    next_epoch();
    bool ok = apply(t, edits, n_edits, new_root, why);
    if (ok)
        ok = mark_from_edits(t, edits, n_edits, why);
    if (!ok)
        undo(t);
    undo_.clear();
    return ok;
}

void ProgramBuilder::build_full(const Topology &t, Program &out)
{
    if ((int32_t)mark_.size() != t.nb)
        resize(t.nb);
    next_epoch();
    for (int32_t v = t.n; v < t.nb; v++)
    {
        mark_[v] = epoch_;
        dirty_list_.push_back(v);
    }
    std::vector<int32_t> tops;
    const int32_t a = t.left[t.root], b = t.right[t.root];
    if (a >= t.n)
        tops.push_back(a);
    if (b >= t.n)
        tops.push_back(b);
    compute_need(t, tops);
    acc_live_ = false;
    depth_ = 0;
    emit_rooted(t, out);
    out.dirty = t.nb - t.n;
}

void ProgramBuilder::build_flagged(const Topology &t, const uint8_t *dirty, Program &out)
{
    if ((int32_t)mark_.size() != t.nb)
        resize(t.nb);
    next_epoch();
    for (int32_t v = t.n; v < t.nb; v++)
        if (dirty[v])
        {
            mark_[v] = epoch_;
            dirty_list_.push_back(v);
        }
    // dirty nodes under a clean parent are evaluated and stored but consumed by nobody:
    // the reference would leave the clean parent's cached set untouched (TreeEvaluation.c:191-202)
    std::vector<int32_t> orphans, tops;
    for (int32_t v : dirty_list_)
    {
        const int32_t p = t.parent[v];
        if (p == t.root)
            tops.push_back(v);
        else if (!is_dirty(p))
            orphans.push_back(v);
    }
    std::vector<int32_t> all(orphans);
    all.insert(all.end(), tops.begin(), tops.end());
    compute_need(t, all);
    acc_live_ = false;
    depth_ = 0;
    for (int32_t v : orphans)
    {
        emit_subtree(t, v, out);
        acc_live_ = false; // result stored, feeds nothing
    }
    emit_rooted(t, out);
    out.dirty = (int32_t)dirty_list_.size();
}

} // namespace lvbgpu
