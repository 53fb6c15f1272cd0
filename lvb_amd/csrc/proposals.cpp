// proposals.cpp - see proposals.hpp
#include "proposals.hpp"

#include <algorithm>

namespace lvbgpu
{

namespace
{

// pending child-pair rewrites layered over an unmodified topology
class Overlay
{
  public:
    Overlay(const Topology &t, std::vector<Edit> &out) : t_(t), out_(out), base_(out.size()) {}
    int32_t left(int32_t v) const
    {
        const int i = find(v);
        return i < 0 ? t_.left[v] : out_[i].left;
    }
    int32_t right(int32_t v) const
    {
        const int i = find(v);
        return i < 0 ? t_.right[v] : out_[i].right;
    }
    void set(int32_t v, int32_t l, int32_t r)
    {
        const int i = find(v);
        if (i < 0)
            out_.push_back({v, l, r});
        else
        {
            out_[i].left = l;
            out_[i].right = r;
        }
    }
    void replace_child(int32_t v, int32_t oldc, int32_t newc)
    {
        const int32_t l = left(v), r = right(v);
        if (l == oldc)
            set(v, newc, r);
        else
            set(v, l, newc);
    }
    int count() const { return (int)(out_.size() - base_); }

  private:
    int find(int32_t v) const
    {
        for (size_t i = base_; i < out_.size(); i++)
            if (out_[i].node == v)
                return (int)i;
        return -1;
    }
    const Topology &t_;
    std::vector<Edit> &out_;
    size_t base_;
};

inline int32_t sister_of(const Topology &t, int32_t v)
{
    const int32_t p = t.parent[v];
    return t.left[p] == v ? t.right[p] : t.left[p];
}

bool is_descendant(const Topology &t, int32_t ancestor, int32_t v)
{
    for (int32_t p = t.parent[v]; p != UNSET; p = t.parent[p])
        if (p == ancestor)
            return true;
    return false;
}

// prune src (with its parent sp) and graft it on the edge above dest; `top` is what hangs under
// the re-used node sp next to dest (src itself for SPR, the new subtree top for TBR)
void prune_and_graft(const Topology &t, Overlay &ov, int32_t src, int32_t dest, int32_t top)
{
    const int32_t sp = t.parent[src];
    const int32_t ss = sister_of(t, src);
    const int32_t pp = t.parent[sp];
    ov.replace_child(pp, sp, ss);               // free the pruned parent (TreeOperations.c:287-299)
    ov.replace_child(t.parent[dest], dest, sp); // make room above dest   (301-316)
    ov.set(sp, dest, top);                      // (317-324)
}

} // namespace

void subtree_leaves(const Topology &t, int32_t top, std::vector<int32_t> &leaves)
{
    std::vector<int32_t> st{top};
    while (!st.empty())
    {
        const int32_t v = st.back();
        st.pop_back();
        if (t.left[v] < 0)
            leaves.push_back(v);
        else
        {
            st.push_back(t.right[v]);
            st.push_back(t.left[v]);
        }
    }
}

bool spr_move_allowed(const Topology &t, int32_t src, int32_t dest)
{
    if (src < 0 || src >= t.nb || dest < 0 || dest >= t.nb)
        return false;
    if (src == t.root || src == t.left[t.root] || src == t.right[t.root])
        return false; // TreeOperations.c:256-259
    if (dest == src || dest == t.parent[src] || dest == sister_of(t, src) || dest == t.root)
        return false; // 268-271
    return !is_descendant(t, src, dest);
}

int nni_edits(const Topology &t, int32_t u, bool swap_right, std::vector<Edit> &out)
{
    Overlay ov(t, out);
    const int32_t v = t.parent[u];
    const int32_t a = t.left[u], b = t.right[u];
    const int32_t c = sister_of(t, u);
    if (swap_right)
    {
        ov.replace_child(v, c, b); // TreeOperations.c:186-194
        ov.set(u, a, c);
    }
    else
    {
        ov.replace_child(v, c, a); // 197-204
        ov.set(u, b, c);
    }
    return ov.count();
}

int spr_edits(const Topology &t, int32_t src, int32_t dest, std::vector<Edit> &out)
{
    Overlay ov(t, out);
    prune_and_graft(t, ov, src, dest, src);
    return ov.count();
}

int tbr_edits(const Topology &t, int32_t src, int32_t dest, int32_t x, std::vector<Edit> &out)
{
    // x: leaf of src's subtree, not a child of src; the subtree is re-rooted on the edge above x.
    // Path P0 = parent(x) .. Pk = src.  P0 becomes the top with children (P1, x); every Pi on the
    // way takes (P(i+1), the sister displaced one level below); src keeps its other child and
    // receives the last displaced sister (TreeOperations.c:455-507).
    Overlay ov(t, out);
    std::vector<int32_t> path;
    for (int32_t p = t.parent[x]; p != src; p = t.parent[p])
        path.push_back(p);
    path.push_back(src);
    const int k = (int)path.size() - 1;
    int32_t displaced = sister_of(t, x);
    ov.set(path[0], path[1], x);
    for (int i = 1; i < k; i++)
    {
        const int32_t pi = path[i];
        const int32_t other = (t.left[pi] == path[i - 1]) ? t.right[pi] : t.left[pi];
        ov.set(pi, path[i + 1], displaced);
        displaced = other;
    }
    ov.replace_child(src, path[k - 1], displaced);
    prune_and_graft(t, ov, src, dest, path[0]);
    return ov.count();
}

static void draw_spr(const Topology &t, Rng &rng, int32_t &src, int32_t &dest)
{
    do
        src = (int32_t)rng.below((uint32_t)t.nb);
    while (src == t.root || src == t.left[t.root] || src == t.right[t.root]);
    do
        dest = (int32_t)rng.below((uint32_t)t.nb);
    while (!spr_move_allowed(t, src, dest));
}

MoveParams draw_move(const Topology &t, int kind, Rng &rng)
{
    if (kind == MOVE_NNI)
    {
        const int32_t u = t.n + (int32_t)rng.below((uint32_t)(t.nb - t.n)); // any internal node
        return {MOVE_NNI, u, rng.uniform() < 0.5 ? 1 : 0, -1};
    }
    int32_t src, dest;
    draw_spr(t, rng, src, dest);
    if (kind == MOVE_SPR)
        return {MOVE_SPR, src, dest, -1};
    std::vector<int32_t> leaves;
    subtree_leaves(t, src, leaves);
    if (leaves.size() <= 2) // nothing to re-root (TreeOperations.c:436)
        return {MOVE_TBR, src, dest, -1};
    int32_t x;
    do
        x = leaves[rng.below((uint32_t)leaves.size())];
    while (x == t.left[src] || x == t.right[src]); // 448-451
    return {MOVE_TBR, src, dest, x};
}

int move_edits(const Topology &t, const MoveParams &m, std::vector<Edit> &out)
{
    if (m.kind == MOVE_NNI)
        return nni_edits(t, m.a, m.b != 0, out);
    if (m.kind == MOVE_SPR || m.c < 0)
        return spr_edits(t, m.a, m.b, out);
    return tbr_edits(t, m.a, m.b, m.c, out);
}

int propose_nni(const Topology &t, Rng &rng, std::vector<Edit> &out) { return move_edits(t, draw_move(t, MOVE_NNI, rng), out); }
int propose_spr(const Topology &t, Rng &rng, std::vector<Edit> &out) { return move_edits(t, draw_move(t, MOVE_SPR, rng), out); }
int propose_tbr(const Topology &t, Rng &rng, std::vector<Edit> &out) { return move_edits(t, draw_move(t, MOVE_TBR, rng), out); }

int propose(const Topology &t, int kind, Rng &rng, std::vector<Edit> &out)
{
    return move_edits(t, draw_move(t, kind < 0 || kind > 2 ? MOVE_TBR : kind, rng), out);
}

int reroot_edits(const Topology &t, int32_t newroot, std::vector<Edit> &out)
{
    Overlay ov(t, out);
    if (newroot == t.root)
        return 0;
    // every node on the way up takes (its old parent, its old sister) as children; the old root
    // becomes an ordinary leaf (TreeOperations.c:598-628)
    for (int32_t c = newroot; c != t.root; c = t.parent[c])
        ov.set(c, t.parent[c], sister_of(t, c));
    ov.set(t.root, UNSET, UNSET);
    return ov.count();
}

void random_topology(int32_t n, Rng &rng, Topology &out)
{
    const int32_t nb = 2 * n - 3;
    std::vector<int32_t> left(nb, UNSET), right(nb, UNSET), parent(nb, UNSET);
    std::vector<int32_t> perm(n - 1);
    for (int32_t i = 0; i < n - 1; i++)
        perm[i] = i + 1;
    for (int32_t i = n - 2; i > 0; i--)
        std::swap(perm[i], perm[rng.below((uint32_t)i + 1)]);
    left[0] = perm[0];
    right[0] = perm[1];
    parent[perm[0]] = parent[perm[1]] = 0;
    std::vector<int32_t> leaves{perm[0], perm[1]};
    int32_t next_internal = n;
    for (int32_t k = 2; k < n - 1; k++)
    {
        const int32_t x = leaves[rng.below((uint32_t)leaves.size())];
        const int32_t in = next_internal++;
        const int32_t p = parent[x];
        (left[p] == x ? left[p] : right[p]) = in;
        parent[in] = p;
        left[in] = x;
        right[in] = perm[k];
        parent[x] = parent[perm[k]] = in;
        leaves.push_back(perm[k]);
    }
    std::string why;
    out.assign(n, left.data(), right.data(), 0, &why);
}

} // namespace lvbgpu
