// host_api.cpp - C ABI of include/lvbhost.h: topology object, proposal generators, program
// introspection.  The SA loop lives in anneal.cpp.  No scoring happens in this library.
#include "../../include/lvbhost.h"

#include <algorithm>
#include <cctype>
#include <climits>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <new>
#include <sstream>
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

// ------------------------------------------------------------------ best-topology set

void BestSet::reset(int32_t n)
{
    key.resize(n);
    Rng r(0xD1B54A32D192ED03ull);
    for (int32_t i = 0; i < n; i++)
        key[i] = r.next();
    clear();
}

uint64_t BestSet::hash(const Topology &t, std::vector<uint64_t> &sub) const
{
    // subtree key = XOR of the keys of its taxa, children before parents (explicit postorder)
    sub.assign(t.nb, 0);
    uint64_t all = 0;
    for (int32_t i = 0; i < t.n; i++)
    {
        sub[i] = key[i];
        all ^= key[i];
    }
    std::vector<int32_t> order;
    order.reserve(t.nb);
    std::vector<int32_t> st{t.left[t.root], t.right[t.root]};
    while (!st.empty())
    {
        const int32_t v = st.back();
        st.pop_back();
        order.push_back(v);
        if (t.left[v] >= 0)
        {
            st.push_back(t.left[v]);
            st.push_back(t.right[v]);
        }
    }
    uint64_t h = 0;
    for (auto it = order.rbegin(); it != order.rend(); ++it)
    {
        const int32_t v = *it;
        if (t.left[v] < 0)
            continue;
        sub[v] = sub[t.left[v]] ^ sub[t.right[v]];
        // one bipartition = two complementary sides: canonicalise on the numerically smaller key
        const uint64_t side = sub[v];
        const uint64_t other = all ^ side;
        const uint64_t canon = side < other ? side : other;
        uint64_t z = canon + 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        h += z ^ (z >> 31);
    }
    return h;
}

// the unrooted topology in a form that does not depend on rooting, node numbers or child order
void BestSet::canonical(const Topology &t, std::vector<int32_t> &out)
{
    // The root leaf R carries its two children itself (the edge to its neighbour is folded into the record): as an
    // undirected tree, node R is that neighbour and taxon R hangs from it as a virtual leaf, numbered nb here.
    const int32_t nb = t.nb, vleaf = nb, R = t.root;
    auto neighbours = [&](int32_t v, int32_t nbr[3]) -> int {
        if (v == vleaf)
        {
            nbr[0] = R;
            return 1;
        }
        int k = 0;
        if (v == R)
            nbr[k++] = vleaf;
        else
            nbr[k++] = t.parent[v];
        if (t.left[v] >= 0)
        {
            nbr[k++] = t.left[v];
            nbr[k++] = t.right[v];
        }
        return k;
    };
    auto label = [&](int32_t v) { return v == vleaf ? R : v; }; // taxon of a leaf
    const int32_t start = R == 0 ? vleaf : 0;                  // the leaf of taxon 0
    // orient away from taxon 0: preorder with the node each was reached from
    std::vector<int32_t> order, from((size_t)nb + 1, -1), minlab((size_t)nb + 1, INT32_MAX);
    order.reserve((size_t)nb + 1);
    std::vector<int32_t> st{start};
    while (!st.empty())
    {
        const int32_t v = st.back();
        st.pop_back();
        order.push_back(v);
        int32_t nbr[3];
        const int k = neighbours(v, nbr);
        for (int i = 0; i < k; i++)
            if (nbr[i] != from[v])
            {
                from[nbr[i]] = v;
                st.push_back(nbr[i]);
            }
    }
    for (auto it = order.rbegin(); it != order.rend(); ++it)
    {
        const int32_t v = *it;
        int32_t nbr[3];
        const int k = neighbours(v, nbr);
        if (k == 1)
            minlab[v] = label(v);
        if (from[v] >= 0)
            minlab[from[v]] = std::min(minlab[from[v]], minlab[v]);
    }
    // preorder again, subtrees by their smallest taxon
    out.clear();
    out.reserve((size_t)nb + 1);
    st.assign(1, start);
    while (!st.empty())
    {
        const int32_t v = st.back();
        st.pop_back();
        int32_t nbr[3], kids[2];
        const int k = neighbours(v, nbr);
        int nk = 0;
        for (int i = 0; i < k; i++)
            if (nbr[i] != from[v])
                kids[nk++] = nbr[i];
        if (k == 1 && v != start)
        {
            out.push_back(label(v));
            continue;
        }
        if (v != start)
            out.push_back(-1);
        if (nk == 2 && minlab[kids[0]] > minlab[kids[1]])
            std::swap(kids[0], kids[1]);
        for (int i = nk - 1; i >= 0; i--)
            st.push_back(kids[i]);
    }
}

// A search that is still descending replaces the set at every improvement (clear + insert): the lone tree of a
// fresh set is stored as it is, and its canonical form and hash are only worked out when a second tree of the same
// length turns up and has to be compared with it.
void BestSet::index_all()
{
    std::vector<uint64_t> sub;
    for (size_t i = 0; i < kept.size(); i++)
        if (kept[i].canon.empty())
        {
            Topology t;
            std::string why;
            if (!t.assign((int32_t)key.size(), kept[i].left.data(), kept[i].right.data(), kept[i].root, &why))
                continue;
            canonical(t, kept[i].canon);
            by_hash.emplace(hash(t, sub), i);
        }
}

bool BestSet::insert(const Topology &t)
{
    if (kept.empty())
    {
        kept.push_back({t.left, t.right, t.root, {}});
        return true;
    }
    index_all();
    std::vector<uint64_t> sub;
    const uint64_t h = hash(t, sub);
    std::vector<int32_t> canon;
    canonical(t, canon);
    const auto range = by_hash.equal_range(h);
    for (auto it = range.first; it != range.second; ++it)
        if (kept[it->second].canon == canon)
            return false; // the same topology, compared exactly
    by_hash.emplace(h, kept.size());
    kept.push_back({t.left, t.right, t.root, std::move(canon)});
    return true;
}

BestSet::Kept BestSet::pop_last()
{
    index_all();
    Kept k = std::move(kept.back());
    kept.pop_back();
    for (auto it = by_hash.begin(); it != by_hash.end(); ++it)
        if (it->second == kept.size())
        {
            by_hash.erase(it);
            break;
        }
    return k;
}

void BestSet::push_kept(Kept &&k, std::vector<uint64_t> &scratch)
{
    (void)scratch;
    k.canon.clear(); // indexed again when needed
    kept.push_back(std::move(k));
}

extern "C" uint64_t lvbhost_tree_topology_hash(const lvbhost_tree *t)
{
    std::vector<uint64_t> sub;
    return t->best.hash(t->topo, sub);
}

extern "C" int32_t lvbhost_tree_canonical(const lvbhost_tree *t, int32_t *out, int32_t cap)
{
    if (!t || !out)
        return LVBGPU_E_ARG;
    std::vector<int32_t> c;
    BestSet::canonical(t->topo, c);
    if ((int32_t)c.size() > cap)
        return LVBGPU_E_ARG;
    memcpy(out, c.data(), c.size() * sizeof(int32_t));
    return (int32_t)c.size();
}

extern "C" int32_t lvbhost_tree_best_count(const lvbhost_tree *t) { return t ? (int32_t)t->best.count() : 0; }
extern "C" int32_t lvbhost_tree_best_kept(const lvbhost_tree *t) { return t ? (int32_t)t->best.kept.size() : 0; }
extern "C" int lvbhost_tree_best_get(const lvbhost_tree *t, int32_t i, int32_t *left, int32_t *right, int32_t *root)
{
    if (!t || i < 0 || i >= (int32_t)t->best.kept.size() || !left || !right || !root)
        return LVBGPU_E_ARG;
    const BestSet::Kept &k = t->best.kept[i];
    memcpy(left, k.left.data(), k.left.size() * 4);
    memcpy(right, k.right.data(), k.right.size() * 4);
    *root = k.root;
    return LVBGPU_OK;
}

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
    t->best.reset(n);
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
    t->best.reset(n);
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
    // (measured: a move costs ~50 ns to draw - handing slices to worker threads costs more than
    // it saves up to B = 16384, so this stays on the calling thread)
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

// ------------------------------------------------------------------ alignment input / tree output

struct lvbhost_alignment
{
    std::vector<std::string> names, rows;
    int64_t m = 0;
};

namespace
{
std::string trimmed(const std::string &s)
{
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a]))
        a++;
    while (b > a && isspace((unsigned char)s[b - 1]))
        b--;
    return s.substr(a, b - a);
}
void append_sequence(std::string &dst, const std::string &chunk)
{
    for (char c : chunk)
        if (!isdigit((unsigned char)c) && !isspace((unsigned char)c))
            dst.push_back((char)toupper((unsigned char)c));
}
} // namespace

extern "C" lvbhost_alignment *lvbhost_alignment_read_phylip(const char *path, char *err, int32_t errcap)
{
    auto fail = [&](const std::string &msg) -> lvbhost_alignment * {
        if (err && errcap > 0)
            snprintf(err, (size_t)errcap, "%s", msg.c_str());
        return nullptr;
    };
    std::ifstream in(path);
    if (!in)
        return fail(std::string("Failed to open alignment file: ") + path);
    const int name_field = 10;
    lvbhost_alignment *a = new (std::nothrow) lvbhost_alignment();
    if (!a)
        return fail("out of memory");
    long n = 0, m = 0;
    std::string line;
    bool started = false;
    std::vector<std::string> lines; // non-empty lines after the header, from the first sequence line on
    while (std::getline(in, line))
    {
        if (!line.empty() && line.back() == '\r')
            line.pop_back();
        if (trimmed(line).empty())
            continue;
        if (n == 0)
        {
            if (sscanf(line.c_str(), "%ld%ld", &n, &m) != 2 || n < 1 || m < 1)
            {
                delete a;
                return fail("Some problem reading the file. Please, check the file format.");
            }
            continue;
        }
        if (!started)
        {
            // the reference skips everything up to the first line that ends in a sequence character
            // (MSAInput.cpp:334-340; its PHYLIP character set has no J, O, U and no digits)
            static const std::string seq_chars = "ABCDEFGHIKLMNPQRSTVWXYZabcdefghiklmnpqrstvwxyz*?-";
            const std::string tl = trimmed(line);
            if (seq_chars.find(tl.back()) == std::string::npos)
                continue;
            started = true;
        }
        lines.push_back(line);
    }
    if (n == 0 || (long)lines.size() < n)
    {
        const std::string msg = "The file has a different number of sequences.\nRead: " + std::to_string(lines.size()) +
                                "\nIn the header: " + std::to_string(n);
        delete a;
        return fail(msg);
    }
    // Two layouts share this syntax: interleaved (blocks of n lines, names in the first block) and
    // sequential (each taxon's lines together, name on its first line).  Assemble both readings
    // and keep the one in which every sequence has exactly m sites (interleaved wins a tie).
    auto assemble = [&](bool interleaved, std::vector<std::string> &names, std::vector<std::string> &rows) -> bool {
        names.clear();
        rows.clear();
        const long total = (long)lines.size();
        if (total % n != 0)
            return false;
        const long per = total / n;
        for (long i = 0; i < n; i++)
        {
            const std::string &first = lines[(size_t)(interleaved ? i : i * per)];
            if ((int)first.size() < name_field)
                return false;
            names.push_back(trimmed(first.substr(0, name_field)));
            rows.emplace_back();
            append_sequence(rows.back(), first.substr(name_field));
            for (long k = 1; k < per; k++)
                append_sequence(rows.back(), lines[(size_t)(interleaved ? k * n + i : i * per + k)]);
            if ((long)rows.back().size() != m)
                return false;
        }
        return true;
    };
    if (!assemble(true, a->names, a->rows) && !assemble(false, a->names, a->rows))
    {
        // report what the interleaved reading found, as the reference's message does
        std::vector<std::string> nm, rw;
        assemble(true, nm, rw);
        std::string msg = "Some problem reading the file. Please, check the file format.";
        if (!rw.empty())
            msg = "This sequence " + nm.back() + " has a different length " + std::to_string(rw.back().size()) +
                  " from the one read in the header: " + std::to_string(m);
        delete a;
        return fail(msg);
    }
    a->m = m;
    return a;
}

// ---- FASTA / NEXUS / CLUSTAL (the reference's -f 1/2/3).  Own design, not the reference's line-position state
// machines: every format is reduced to a stream of (name, chunk) records and the chunks are appended to the
// taxon of that NAME, in order of first appearance - so wrapped, interleaved and multi-block files need no
// bookkeeping of "which line of the block is this".  What the reference's reader accepts is accepted here with the
// same result (tests/test_io_vs_reference.py compares both on its example in every format and on wrapped /
// interleaved variants); the error texts are ours.

namespace
{
struct NamedRows
{
    lvbhost_alignment &a;
    std::map<std::string, size_t> index;
    void add(const std::string &name, const std::string &chunk)
    {
        auto it = index.find(name);
        if (it == index.end())
        {
            it = index.emplace(name, a.names.size()).first;
            a.names.push_back(name);
            a.rows.emplace_back();
        }
        for (char c : chunk)
            if (!isspace((unsigned char)c))
                a.rows[it->second].push_back(c);
    }
};

std::vector<std::string> words_of(const std::string &line)
{
    std::vector<std::string> w;
    std::istringstream ss(line);
    for (std::string x; ss >> x;)
        w.push_back(x);
    return w;
}
// all words but the last, blank-joined: a taxon label may hold blanks
std::string label_of(const std::vector<std::string> &w, size_t nchunk_words)
{
    std::string out;
    for (size_t i = 0; i + nchunk_words < w.size(); i++)
        out += (i ? " " : "") + w[i];
    return out;
}
std::string lower(std::string s)
{
    for (char &c : s)
        c = (char)tolower((unsigned char)c);
    return s;
}
bool all_digits(const std::string &s) { return !s.empty() && std::all_of(s.begin(), s.end(), [](char c) { return isdigit((unsigned char)c); }); }

// ">name" opens a record; every other line belongs to the open record (records are positional: two may share a label)
bool parse_fasta(const std::vector<std::string> &lines, NamedRows &out, std::string &)
{
    for (const std::string &raw : lines)
    {
        const std::string line = trimmed(raw);
        if (line.empty())
            continue;
        if (line[0] == '>')
        {
            out.a.names.push_back(line.substr(1));
            out.a.rows.emplace_back();
        }
        else if (!out.a.rows.empty())
            for (char c : line)
                if (!isspace((unsigned char)c))
                    out.a.rows.back().push_back(c);
    }
    return true;
}

// "dimensions ntax=.. nchar=..;" gives the expected shape; between "matrix" and the closing ";" every line is
// "<label> <chunk>" (interleaved blocks repeat the labels)
bool parse_nexus(const std::vector<std::string> &lines, NamedRows &out, std::string &why)
{
    long ntax = 0, nchar = 0;
    bool in_matrix = false;
    for (const std::string &raw : lines)
    {
        const std::string line = trimmed(raw), low = lower(line);
        if (line.empty())
            continue;
        if (!in_matrix)
        {
            if (low.compare(0, 10, "dimensions") == 0)
                for (const char *key : {"ntax=", "nchar="})
                {
                    const size_t at = low.find(key);
                    if (at != std::string::npos)
                        (key[1] == 't' ? ntax : nchar) = strtol(low.c_str() + at + strlen(key), nullptr, 10);
                }
            else if (low == "matrix" || low.compare(0, 7, "matrix ") == 0)
                in_matrix = true;
            continue;
        }
        if (line.find(';') != std::string::npos)
        {
            in_matrix = false;
            continue;
        }
        const std::vector<std::string> w = words_of(line);
        if (w.size() >= 2)
            out.add(label_of(w, 1), w.back());
    }
    if (ntax <= 0 || nchar <= 0)
    {
        why = "NEXUS: no 'dimensions ntax=<n> nchar=<m>;' statement found before the matrix; check the file format";
        return false;
    }
    if ((long)out.a.names.size() != ntax)
    {
        why = "NEXUS: the matrix holds a different number of sequences (" + std::to_string(out.a.names.size()) +
              ") than its dimensions statement announces (" + std::to_string(ntax) + ")";
        return false;
    }
    for (size_t i = 0; i < out.a.rows.size(); i++)
        if ((long)out.a.rows[i].size() != nchar)
        {
            why = "NEXUS: sequence '" + out.a.names[i] + "' has a different length (" + std::to_string(out.a.rows[i].size()) +
                  ") than nchar=" + std::to_string(nchar);
            return false;
        }
    return true;
}

// after the "CLUSTAL ..." line: "<label> <chunk> [running count]"; lines that start with a blank are the
// conservation marks under a block
bool parse_clustal(const std::vector<std::string> &lines, NamedRows &out, std::string &why)
{
    bool header = false;
    for (const std::string &line : lines)
    {
        if (!header)
        {
            header = line.find("CLUSTAL") != std::string::npos;
            continue;
        }
        if (line.empty() || isspace((unsigned char)line[0]))
            continue;
        std::vector<std::string> w = words_of(line);
        if (w.size() >= 3 && all_digits(w.back()))
            w.pop_back();
        if (w.size() >= 2)
            out.add(label_of(w, 1), w.back());
    }
    if (!header)
    {
        why = "CLUSTAL: the file does not start with a CLUSTAL header line; check the file format";
        return false;
    }
    return true;
}

// what every format must satisfy before matchange sees it (the reference applies the same conditions after
// reading, MSAInput.cpp:780-849): two or more sequences, one length, upper case, DNAToBinary's alphabet plus O
bool validate_alignment(lvbhost_alignment &a, const std::string &path, std::string &why)
{
    if (a.rows.size() < 2)
    {
        why = (a.rows.empty() ? "Zero sequences were read from " : "Only one sequence was read from ") + path;
        return false;
    }
    for (size_t i = 1; i < a.rows.size(); i++)
        if (a.rows[i].size() != a.rows[0].size())
        {
            why = "The sequence lengths are different in " + path + ": '" + a.names[i] + "' has " + std::to_string(a.rows[i].size()) +
                  " sites, '" + a.names[0] + "' has " + std::to_string(a.rows[0].size());
            return false;
        }
    for (size_t i = 0; i < a.rows.size(); i++)
        for (char &c : a.rows[i])
        {
            c = (char)toupper((unsigned char)c);
            if (!strchr("ACGTUYRWSKMBDHVNX?O-", c))
            {
                why = std::string("This char is not allowed (") + c + ") in sequence '" + a.names[i] + "' of " + path;
                return false;
            }
        }
    a.m = (int64_t)a.rows[0].size();
    return true;
}
} // namespace

extern "C" lvbhost_alignment *lvbhost_alignment_read(const char *path, int format, char *err, int32_t errcap)
{
    auto fail = [&](const std::string &msg) -> lvbhost_alignment * {
        if (err && errcap > 0)
            snprintf(err, (size_t)errcap, "%s", msg.c_str());
        return nullptr;
    };
    if (!path)
        return fail("no file name");
    if (format < 0 || format > 3)
        return fail(std::string("unknown alignment format code for ") + path);
    lvbhost_alignment *a = nullptr;
    std::string why;
    if (format == 0)
    {
        a = lvbhost_alignment_read_phylip(path, err, errcap);
        if (!a)
            return nullptr;
    }
    else
    {
        std::ifstream in(path);
        if (!in)
            return fail(std::string("Failed to open alignment file: ") + path);
        std::vector<std::string> lines;
        for (std::string line; std::getline(in, line);)
        {
            if (!line.empty() && line.back() == '\r')
                line.pop_back();
            lines.push_back(line);
        }
        a = new (std::nothrow) lvbhost_alignment();
        if (!a)
            return fail("out of memory");
        NamedRows out{*a, {}};
        const bool ok = format == 1 ? parse_fasta(lines, out, why) : (format == 2 ? parse_nexus(lines, out, why) : parse_clustal(lines, out, why));
        if (!ok)
        {
            delete a;
            return fail(why);
        }
    }
    if (!validate_alignment(*a, path, why))
    {
        delete a;
        return fail(why);
    }
    return a;
}

extern "C" void lvbhost_alignment_free(lvbhost_alignment *a) { delete a; }
extern "C" int64_t lvbhost_alignment_n(const lvbhost_alignment *a) { return a ? (int64_t)a->rows.size() : 0; }
extern "C" int64_t lvbhost_alignment_m(const lvbhost_alignment *a) { return a ? a->m : 0; }
extern "C" const char *lvbhost_alignment_row(const lvbhost_alignment *a, int64_t i) { return a->rows[(size_t)i].c_str(); }
extern "C" const char *lvbhost_alignment_name(const lvbhost_alignment *a, int64_t i)
{
    return a->names[(size_t)i].c_str();
}

extern "C" int64_t lvbhost_tree_newick(const lvbhost_tree *t, const char *const *names, char *out, int64_t cap)
{
    if (!t || !names || !out || cap < 4)
        return LVBGPU_E_ARG;
    const Topology &tp = t->topo;
    std::string s;
    auto name_of = [&](int32_t v) {
        std::string nm = names[v];
        while (!nm.empty() && nm.back() == ' ')
            nm.pop_back();
        return nm;
    };
    // "(root" then the two subtrees, comma-separated, ")" - explicit stack instead of recursion
    s += "(" + name_of(tp.root);
    struct Item
    {
        int32_t v;
        int stage;
    };
    std::vector<Item> st;
    st.push_back({tp.right[tp.root], 0});
    st.push_back({tp.left[tp.root], 0});
    while (!st.empty())
    {
        Item it = st.back();
        st.pop_back();
        if (it.stage == 1)
        {
            s += ")";
            continue;
        }
        if (s.back() != '(')
            s += ",";
        if (it.v < tp.n)
            s += name_of(it.v);
        else
        {
            s += "(";
            st.push_back({it.v, 1});
            st.push_back({tp.right[it.v], 0});
            st.push_back({tp.left[it.v], 0});
        }
    }
    s += ");\n";
    if ((int64_t)s.size() + 1 > cap)
        return LVBGPU_E_ARG;
    memcpy(out, s.c_str(), s.size() + 1);
    return (int64_t)s.size();
}
