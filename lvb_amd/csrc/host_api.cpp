// host_api.cpp - C ABI of include/lvbhost.h: topology object, proposal generators, program
// introspection.  The SA loop lives in anneal.cpp.  No scoring happens in this library.
#include "../../include/lvbhost.h"

#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "host_tree.hpp"

using namespace lvbgpu;

static_assert(sizeof(lvbgpu_edit) == sizeof(Edit), "edit layout");

namespace
{
int copy_out(const std::vector<Edit> &v, lvbgpu_edit *edits, int32_t cap)
{
    if ((int32_t)v.size() > cap)
        return LVBGPU_E_ARG;
    memcpy(edits, v.data(), v.size() * sizeof(Edit));
    return (int)v.size();
}
} // namespace

extern "C" lvbhost_tree *lvbhost_tree_random(int32_t n, uint64_t seed)
{
    if (n < 3)
        return nullptr;
    lvbhost_tree *t = new (std::nothrow) lvbhost_tree();
    if (!t)
        return nullptr;
    t->rng = Rng(seed);
    random_topology(n, t->rng, t->topo);
    t->pb.resize(t->topo.nb);
    return t;
}

extern "C" lvbhost_tree *lvbhost_tree_from_arrays(int32_t n, const int32_t *left, const int32_t *right, int32_t root,
                                                  uint64_t seed)
{
    if (n < 3 || !left || !right)
        return nullptr;
    lvbhost_tree *t = new (std::nothrow) lvbhost_tree();
    if (!t)
        return nullptr;
    t->rng = Rng(seed);
    std::string why;
    if (!t->topo.assign(n, left, right, root, &why))
    {
        delete t;
        return nullptr;
    }
    t->pb.resize(t->topo.nb);
    return t;
}

extern "C" void lvbhost_tree_free(lvbhost_tree *t) { delete t; }
extern "C" int32_t lvbhost_tree_n(const lvbhost_tree *t) { return t ? t->topo.n : 0; }
extern "C" int32_t lvbhost_tree_root(const lvbhost_tree *t) { return t ? t->topo.root : -1; }

extern "C" void lvbhost_tree_arrays(const lvbhost_tree *t, int32_t *parent, int32_t *left, int32_t *right)
{
    const size_t bytes = (size_t)t->topo.nb * 4;
    if (parent)
        memcpy(parent, t->topo.parent.data(), bytes);
    if (left)
        memcpy(left, t->topo.left.data(), bytes);
    if (right)
        memcpy(right, t->topo.right.data(), bytes);
}

extern "C" void lvbhost_tree_reseed(lvbhost_tree *t, uint64_t seed) { t->rng = Rng(seed); }

extern "C" int lvbhost_propose(lvbhost_tree *t, int kind, lvbgpu_edit *edits, int32_t cap)
{
    if (!t || !edits || kind < 0 || kind > 2 || t->topo.n < 5)
        return LVBGPU_E_ARG;
    t->scratch.clear();
    propose(t->topo, kind, t->rng, t->scratch);
    return copy_out(t->scratch, edits, cap);
}

extern "C" int lvbhost_propose_batch(lvbhost_tree *t, int kind, int32_t B, int32_t *edit_offsets, lvbgpu_edit *edits,
                                     int32_t cap)
{
    if (!t || !edits || !edit_offsets || B < 1 || kind < -1 || kind > 2 || t->topo.n < 5)
        return LVBGPU_E_ARG;
    t->scratch.clear();
    edit_offsets[0] = 0;
    for (int32_t b = 0; b < B; b++)
    {
        propose(t->topo, kind < 0 ? b % 3 : kind, t->rng, t->scratch);
        edit_offsets[b + 1] = (int32_t)t->scratch.size();
    }
    return copy_out(t->scratch, edits, cap);
}

extern "C" int lvbhost_nni_edits(const lvbhost_tree *t, int32_t u, int swap_right, lvbgpu_edit *edits, int32_t cap)
{
    if (!t || !edits || u < t->topo.n || u >= t->topo.nb)
        return LVBGPU_E_ARG;
    std::vector<Edit> v;
    nni_edits(t->topo, u, swap_right != 0, v);
    return copy_out(v, edits, cap);
}

extern "C" int lvbhost_spr_edits(const lvbhost_tree *t, int32_t src, int32_t dest, lvbgpu_edit *edits, int32_t cap)
{
    if (!t || !edits || !spr_move_allowed(t->topo, src, dest))
        return LVBGPU_E_ARG;
    std::vector<Edit> v;
    spr_edits(t->topo, src, dest, v);
    return copy_out(v, edits, cap);
}

extern "C" int lvbhost_tbr_edits(const lvbhost_tree *t, int32_t src, int32_t dest, int32_t x, lvbgpu_edit *edits,
                                 int32_t cap)
{
    if (!t || !edits || !spr_move_allowed(t->topo, src, dest) || x < 0 || x >= t->topo.n)
        return LVBGPU_E_ARG;
    const Topology &tp = t->topo;
    if (x == tp.left[src] || x == tp.right[src])
        return LVBGPU_E_ARG;
    bool below = false;
    for (int32_t p = tp.parent[x]; p != UNSET; p = tp.parent[p])
        if (p == src)
            below = true;
    if (!below)
        return LVBGPU_E_ARG;
    std::vector<Edit> v;
    tbr_edits(tp, src, dest, x, v);
    return copy_out(v, edits, cap);
}

extern "C" int lvbhost_reroot_edits(const lvbhost_tree *t, int32_t newroot, lvbgpu_edit *edits, int32_t cap)
{
    if (!t || !edits || newroot < 0 || newroot >= t->topo.n)
        return LVBGPU_E_ARG;
    std::vector<Edit> v;
    reroot_edits(t->topo, newroot, v);
    return copy_out(v, edits, cap);
}

extern "C" int lvbhost_tree_apply(lvbhost_tree *t, const lvbgpu_edit *edits, int32_t n_edits, int32_t new_root)
{
    if (!t || n_edits < 0 || (n_edits && !edits))
        return LVBGPU_E_ARG;
    std::string why;
    if (!t->pb.apply_edits(t->topo, reinterpret_cast<const Edit *>(edits), n_edits, new_root, &why))
        return LVBGPU_E_TOPOLOGY;
    return LVBGPU_OK;
}

extern "C" int lvbhost_program(const lvbhost_tree *tc, int mode, const lvbgpu_edit *edits, int32_t n_edits,
                               int32_t new_root, const uint8_t *dirty_flags, uint32_t *toks, int32_t tok_cap,
                               int32_t *ntok, int32_t *dsts, int32_t dst_cap, int32_t *ndst, int32_t *max_stack,
                               int32_t *n_dirty)
{
    if (!tc || !toks || !dsts || !ntok || !ndst)
        return LVBGPU_E_ARG;
    lvbhost_tree *t = const_cast<lvbhost_tree *>(tc); // builder scratch; topology is restored
    Program prog;
    std::string why;
    if (mode == 0)
    {
        if (!t->pb.build_candidate(t->topo, reinterpret_cast<const Edit *>(edits), n_edits, new_root, prog, &why))
            return LVBGPU_E_TOPOLOGY;
    }
    else if (mode == 1)
        t->pb.build_full(t->topo, prog);
    else if (mode == 2 && dirty_flags)
        t->pb.build_flagged(t->topo, dirty_flags, prog);
    else
        return LVBGPU_E_ARG;
    if ((int32_t)prog.toks.size() > tok_cap || (int32_t)prog.dsts.size() > dst_cap)
        return LVBGPU_E_ARG;
    memcpy(toks, prog.toks.data(), prog.toks.size() * 4);
    memcpy(dsts, prog.dsts.data(), prog.dsts.size() * 4);
    *ntok = (int32_t)prog.toks.size();
    *ndst = (int32_t)prog.dsts.size();
    if (max_stack)
        *max_stack = prog.max_stack;
    if (n_dirty)
        *n_dirty = prog.dirty;
    return LVBGPU_OK;
}

extern "C" int lvbhost_tree_upload(lvbgpu_ctx *ctx, const lvbhost_tree *t, int64_t *length_out)
{
    if (!ctx || !t)
        return LVBGPU_E_ARG;
    return lvbgpu_set_tree(ctx, t->topo.left.data(), t->topo.right.data(), t->topo.root, length_out);
}

// ------------------------------------------------------------------ alignment preparation

extern "C" int64_t lvbhost_variable_columns(int64_t n, int64_t m, const char *const *rows, uint8_t *keep)
{
    // a column survives iff some row differs from row 0 in it, compared as raw characters
    // (reference constchar, DataOperations.c:283-296; note its `togo` flag means KEEP)
    if (!rows || !keep || n < 1 || m < 0)
        return LVBGPU_E_ARG;
    int64_t kept = 0;
    for (int64_t k = 0; k < m; k++)
    {
        const char c0 = rows[0][k];
        uint8_t varies = 0;
        for (int64_t i = 1; i < n && !varies; i++)
            varies = rows[i][k] != c0;
        keep[k] = varies;
        kept += varies;
    }
    return kept;
}

extern "C" int64_t lvbhost_min_tree_length(int64_t n, int64_t m, const char *const *rows)
{
    if (!rows || n < 1 || m < 0)
        return LVBGPU_E_ARG;
    int64_t total = 0;
    for (int64_t k = 0; k < m; k++)
    {
        char seen[8];
        int nseen = 0;
        bool over = false;
        for (int64_t i = 0; i < n && !over; i++)
        {
            const char c = rows[i][k];
            bool known = false;
            for (int s = 0; s < nseen; s++)
                known |= seen[s] == c;
            if (known)
                continue;
            if (c != '-' && c != '?' && c != 'N' && c != 'X')
                seen[nseen++] = c;
            over = nseen > 5; // MAXSTATES, LVB.h:109
        }
        total += over ? 5 : (int64_t)nseen - 1;
    }
    return total;
}
