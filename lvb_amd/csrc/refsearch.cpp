// refsearch.cpp - the reference's own search trajectory, reproduced decision for decision on the
// device scorer.
//
// Same seed => same random start trees (PullRandomTree, TreeOperations.c:799-811, 829-912,
// 957-1059), same starting temperature (StartingTemperature.c:49-195), same sequence of proposals
// (mutate_nni/spr/tbr TreeOperations.c:160-541 drawing from RandomNumberGenerator.c's stream),
// same accept/reject decisions, cooling steps, re-roots and treestack contents (Solve.c:144-479,
// Treestack.c:231-306), hence the same rearrangement count, score and output trees as the
// reference program - with every tree length computed by lvbgpu_* (HIP), never here.
//
// How a serial chain is fed to a batched scorer without changing it: a REJECTED proposal always
// costs the reference exactly one uni() after its move draws (Solve.c:363 / 368), and while
// nothing is accepted the loop's own state (iteration parity, proposal counts, temperature, move
// probabilities) evolves in a way that does not depend on the lengths.  So B proposals are drawn
// ahead under the assumption "all rejected", scored in one device step, and then consumed in
// order with the real lengths.  At the first accepted one the random stream is rewound to the
// snapshot taken after that proposal's own draws (plus the acceptance draw if it was a worse
// tree), the tree is committed, and the rest of the batch is dropped.  A batch never crosses a
// re-root tick (Solve.c:240: every 1000th iteration) because that changes the tree's numbering.
#include "../../include/lvbhost.h"

#include <algorithm>
#include <cstdlib>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "host_tree.hpp"
#include "refrng.hpp"

using namespace lvbgpu;

struct lvbhost_refrng
{
    Uni g;
};

namespace
{

constexpr double LVB_EPS = 1e-11;          // LVB.h:102
constexpr double FROZEN_T = 0.0001;        // LVB.h:112
constexpr int64_t REROOT_INTERVAL = 1000;  // LVB.h:99
constexpr double DBL_EPS = 2.220446049250313e-16;

using Clock = std::chrono::steady_clock;
inline double since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

// ---- proposals drawn the reference's way ----------------------------------------------------

// A move as its parameters (lvbgpu_move): NNI a = u, b = 1 if u's right child is given away; SPR
// a = src, b = dest; TBR a = src, b = dest, c = the leaf the subtree is re-rooted at (-1: as SPR).
// Drawing consumes the stream exactly as the reference's generators do; turning a move into edits
// (or handing it to the device, lvbgpu_score_moves) draws nothing.
using Move = lvbgpu_move;

// mutate_nni (TreeOperations.c:174, 184): one randpint for the internal branch, one uni for the side
Move draw_nni(const Topology &t, Uni &g)
{
    const int32_t u = (int32_t)g.randpint(t.nb - t.n - 1) + t.n;
    return Move{MOVE_NNI, u, g.uni() < 0.5 ? 1 : 0, -1};
}

// the two rejection loops shared by mutate_spr and mutate_tbr (256-271 / 367-382)
void draw_src_dest(const Topology &t, Uni &g, int32_t &src, int32_t &dest)
{
    do
        src = (int32_t)g.randpint(t.nb - 1);
    while (src == t.root || src == t.left[t.root] || src == t.right[t.root]);
    do
        dest = (int32_t)g.randpint(t.nb - 1);
    while (!spr_move_allowed(t, src, dest));
}

Move draw_spr(const Topology &t, Uni &g)
{
    int32_t src, dest;
    draw_src_dest(t, g, src, dest);
    return Move{MOVE_SPR, src, dest, -1};
}

// mutate_tbr: as SPR, then (subtrees of more than two leaves only) re-root the moved subtree at a
// random leaf that is not a child of its top (439-461)
Move draw_tbr(const Topology &t, Uni &g, std::vector<int32_t> &leaves)
{
    int32_t src, dest;
    draw_src_dest(t, g, src, dest);
    leaves.clear();
    subtree_leaves(t, src, leaves);
    const int64_t size = (int64_t)leaves.size();
    if (size <= 2)
        return Move{MOVE_TBR, src, dest, -1};
    int32_t x;
    do
        x = leaves[(size_t)g.randpint(size - 1)];
    while (x == t.left[src] || x == t.right[src]);
    return Move{MOVE_TBR, src, dest, x};
}

Move draw_move_params(const Topology &t, int kind, Uni &g, std::vector<int32_t> &scratch)
{
    switch (kind)
    {
    case MOVE_NNI: return draw_nni(t, g);
    case MOVE_SPR: return draw_spr(t, g);
    default: return draw_tbr(t, g, scratch);
    }
}

int move_edits(const Topology &t, const Move &m, std::vector<Edit> &out)
{
    if (m.kind == MOVE_NNI)
        return nni_edits(t, m.a, m.b != 0, out);
    if (m.kind == MOVE_SPR || m.c < 0)
        return spr_edits(t, m.a, m.b, out);
    return tbr_edits(t, m.a, m.b, m.c, out);
}

int draw_move(const Topology &t, int kind, Uni &g, std::vector<Edit> &out, std::vector<int32_t> &scratch)
{
    return move_edits(t, draw_move_params(t, kind, g, scratch), out);
}

// arbreroot (TreeOperations.c:639-656)
int32_t draw_new_root(const Topology &t, Uni &g)
{
    int32_t nr;
    do
        nr = (int32_t)g.randpint(t.n - 1);
    while (nr == t.root);
    return nr;
}

// ---- PullRandomTree -----------------------------------------------------------------------

// exchange the labels of records a and b everywhere (what one swap of tree_make_canonical,
// TreeOperations.c:852-877, amounts to)
void swap_labels(std::vector<int32_t> &parent, std::vector<int32_t> &left, std::vector<int32_t> &right, int32_t a,
                 int32_t b)
{
    std::swap(parent[a], parent[b]);
    std::swap(left[a], left[b]);
    std::swap(right[a], right[b]);
    auto relabel = [a, b](int32_t &f) {
        if (f == a)
            f = b;
        else if (f == b)
            f = a;
    };
    for (size_t r = 0; r < parent.size(); r++)
    {
        relabel(parent[r]);
        relabel(left[r]);
        relabel(right[r]);
    }
}

bool random_tree(int32_t n, Uni &g, Topology &out, std::string *why)
{
    const int32_t nb = 2 * n - 3;
    std::vector<int32_t> parent(nb, UNSET), left(nb, UNSET), right(nb, UNSET);
    std::vector<uint8_t> isleaf(nb, 0);
    // GenerateRandomTopology (957-1011): root 0 with leaves 1 and 2, then leaves sprout pairs
    left[0] = 1;
    right[0] = 2;
    parent[1] = parent[2] = 0;
    isleaf[0] = isleaf[1] = isleaf[2] = 1;
    int32_t nextfree = 3;
    for (int32_t leaves = 3; leaves < n; leaves++)
    {
        int32_t grow;
        do
            grow = 1 + (int32_t)g.randpint(nextfree - 2);
        while (!isleaf[grow]);
        left[grow] = nextfree;
        parent[nextfree] = grow;
        isleaf[nextfree++] = 1;
        right[grow] = nextfree;
        parent[nextfree] = grow;
        isleaf[nextfree++] = 1;
        isleaf[grow] = 0;
    }
    // randleaf (1013-1059): taxa to leaves in record order, rejection-sampled
    std::vector<int32_t> taxon(nb, UNSET);
    std::vector<uint8_t> used(n, 0);
    for (int32_t i = 0; i < nb; i++)
        if (isleaf[i])
        {
            int32_t cand;
            do
                cand = (int32_t)g.randpint(n - 1);
            while (used[cand]);
            taxon[i] = cand;
            used[cand] = 1;
        }
    // tree_make_canonical (829-912): swap records until taxon k sits in record k ...
    bool swapped;
    do
    {
        swapped = false;
        for (int32_t i = 0; i < nb; i++)
        {
            const int32_t o = taxon[i];
            if (o != UNSET && o != i)
            {
                swap_labels(parent, left, right, i, o);
                std::swap(taxon[i], taxon[o]);
                swapped = true;
            }
        }
    } while (swapped);
    int32_t root = UNSET;
    for (int32_t i = 0; i < n; i++)
        if (parent[i] == UNSET)
            root = i;
    if (!out.assign(n, left.data(), right.data(), root, why))
        return false;
    // ... and put the root at taxon 0 (897-900)
    if (root != 0)
    {
        std::vector<Edit> ed;
        reroot_edits(out, 0, ed);
        ProgramBuilder pb(nb);
        if (!pb.apply_edits(out, ed.data(), (int32_t)ed.size(), 0, why))
            return false;
    }
    return true;
}

// ---- the serial loop's own state (everything of Anneal() that is not a tree or a length) -----

struct LoopState
{
    int algorithm = 1, cooling = 0;
    int64_t maxaccept = 5, maxpropose = 2000, maxfail = 40;
    double t0 = 0.0, t = 0.0;
    int64_t accepted = 0, failedcnt = 0, iter = 0, proposed = 0, t_n = 0, current_iter = 0;
    double counter[3] = {1, 1, 1}; // trops_counter (Solve.c:209)
    double prob[3] = {0, 0, 0};    // trops_probs   (210): all moves are TBR until the first cooling step
    int64_t trop = 0;
    double log_eps = 0.0, log_grad = 0.0, log_t0 = 0.0;

    void start(double t_start)
    {
        t0 = t = t_start;
        log_eps = std::log(LVB_EPS);
        log_grad = std::log(0.99);
        log_t0 = std::log(t_start);
    }
    bool draws_for_kind() const { return algorithm >= 1; }
    // Solve.c:251-298; rv is the uni() drawn at 263 (ignored for -a 0)
    int select(double rv)
    {
        if (algorithm == 2)
        {
            const long total = (long)(counter[0] + counter[1] + counter[2]); // a long in the reference (211, 255)
            prob[0] = counter[0] / total;
            prob[1] = counter[1] / total;
            prob[2] = counter[2] / total;
        }
        if (algorithm >= 1)
        {
            const int k = rv < prob[0] ? MOVE_NNI : (rv < prob[0] + prob[1] ? MOVE_SPR : MOVE_TBR);
            if (algorithm == 2)
                trop = k;
            return k;
        }
        return (iter & 1) ? MOVE_SPR : MOVE_NNI;
    }
    // Solve.c:379-466 for one finished iteration; new_best_topology: the accepted tree tied or beat
    // the best and was new to the treestack (316-319).  Returns true when the system is frozen.
    bool finish(bool new_best_topology)
    {
        if (new_best_topology)
            accepted++;
        proposed++;
        bool dect = false;
        if (accepted >= maxaccept)
        {
            failedcnt = 0;
            dect = true;
        }
        else if (proposed >= maxpropose)
        {
            failedcnt++;
            if (failedcnt >= maxfail && t < FROZEN_T)
                return true;
            dect = true;
        }
        if (dect)
        {
            t_n++;
            if (cooling == 0)
            {
                const double ln_t = ((double)t_n) * log_grad + log_t0;
                if (ln_t < log_eps)
                    t = LVB_EPS;
                else
                    t = std::pow(0.99, (double)t_n) * t0;
                if (algorithm == 1)
                {
                    prob[2] = t / t0;
                    prob[1] = (1 - prob[2]) / 2;
                    prob[0] = prob[1];
                }
            }
            else
            {
                t = t0 - (10 * LVB_EPS) * t_n;
                if (t < DBL_EPS || t <= LVB_EPS)
                    t = LVB_EPS;
            }
            proposed = 0;
            accepted = 0;
        }
        iter++;
        if (algorithm == 2)
        {
            // changeAcc is only ever set for -a 1 (Solve.c:330, 374), so with -a 2 the two moves
            // that were not tried always gain half a count (452-466)
            for (int i = 0; i < 3; i++)
                if (trop != i)
                    counter[i] += 0.5;
        }
        return false;
    }
};

// Metropolis rule of Solve.c:303-378 / StartingTemperature.c:131-166.  `worse_draw` is the uni()
// consumed when the proposal is longer.
struct Verdict
{
    bool accept, worse;
};
inline double energy_step(double min_len, int64_t cur, int64_t prop)
{
    double deltah = (min_len / (double)cur) - (min_len / (double)prop);
    if (deltah > 1.0)
        deltah = 1.0;
    return deltah;
}
inline bool accept_worse(double deltah, double t, double log_eps, double draw)
{
    if (-deltah < t * log_eps)
        return false; // pacc taken as 0; the draw is consumed all the same
    return draw < std::exp(-deltah / t);
}

// ---- the speculative driver -------------------------------------------------------------------

struct Driver
{
    lvbgpu_ctx *ctx = nullptr;
    Topology topo;
    ProgramBuilder pb;
    Uni rng;
    int64_t cur_len = 0;
    double min_len = 0.0;
    int32_t max_batch = 512;
    double run_len = 4.0; // running mean of "proposals consumed per device step"

    struct Cand
    {
        int kind;
        double rv;   // the draw that chose the kind (-a 1, -a 2)
        Uni after;   // stream state after this proposal's own draws
    };
    std::vector<Cand> cands;
    std::vector<Move> moves;  // every proposal of the batch as parameters
    std::vector<Edit> edits;  // ... and, for batches scored from host-built programs, as edits
    std::vector<int32_t> offs, scratch;
    std::vector<int64_t> lens;
    int32_t device_moves_min = 128; // batches at least this long go to lvbgpu_score_moves (< 0: never)
    int64_t device_move_steps = 0;
    // accounting
    int64_t scored = 0, steps = 0, commits = 0, reroots = 0;
    double dev_seconds = 0.0;

    int set_tree()
    {
        const auto t0 = Clock::now();
        const int rc = lvbgpu_set_tree(ctx, topo.left.data(), topo.right.data(), topo.root, &cur_len);
        dev_seconds += since(t0);
        return rc;
    }
    int commit(const Edit *e, int32_t ne, int32_t new_root)
    {
        const auto t0 = Clock::now();
        // asynchronous: the length is known (scored candidate) or unchanged (re-root)
        int rc = lvbgpu_commit(ctx, ne, reinterpret_cast<const lvbgpu_edit *>(e), new_root, nullptr);
        dev_seconds += since(t0);
        std::string why;
        if (rc == LVBGPU_OK && !pb.apply_edits(topo, e, ne, new_root, &why))
            rc = LVBGPU_E_TOPOLOGY;
        commits++;
        return rc;
    }
    int reroot()
    {
        const int32_t nr = draw_new_root(topo, rng);
        std::vector<Edit> ed;
        reroot_edits(topo, nr, ed);
        reroots++;
        return commit(ed.data(), (int32_t)ed.size(), nr);
    }
    void begin_batch()
    {
        cands.clear();
        moves.clear();
    }
    // draw one more proposal of `kind` against the current tree and assume it will be rejected
    void speculate(int kind, double rv)
    {
        moves.push_back(draw_move_params(topo, kind, rng, scratch));
        cands.push_back({kind, rv, rng});
        (void)rng.uni(); // the rejected proposal's acceptance draw
    }
    int score()
    {
        const int32_t B = (int32_t)cands.size();
        lens.resize((size_t)B);
        int rc = LVBGPU_OK;
        bool on_device = device_moves_min >= 0 && B >= device_moves_min;
        if (on_device)
        {
            // long batch: 16 bytes per proposal go to the device, which builds the programs itself
            const auto t0 = Clock::now();
            rc = lvbgpu_score_moves(ctx, B, moves.data(), lens.data());
            dev_seconds += since(t0);
            device_move_steps++;
            // a move the device generator could not represent in its fixed buffers has no length: this run must
            // not guess, so the whole batch goes through the host's program builder instead
            if (rc == LVBGPU_OK && std::find(lens.begin(), lens.end(), INT64_MAX) != lens.end())
                on_device = false;
        }
        if (rc == LVBGPU_OK && !on_device)
        {
            edits.clear();
            offs.assign(1, 0);
            for (const Move &m : moves)
            {
                move_edits(topo, m, edits);
                offs.push_back((int32_t)edits.size());
            }
            const auto t0 = Clock::now();
            rc = lvbgpu_score_batch(ctx, B, offs.data(), reinterpret_cast<const lvbgpu_edit *>(edits.data()), nullptr,
                                    lens.data());
            dev_seconds += since(t0);
        }
        scored += B;
        steps++;
        return rc;
    }
    // decide proposal b exactly as the serial loop would; leaves the stream where the reference's
    // would be after this iteration when the proposal is accepted
    Verdict decide(int32_t b, double t, double log_eps)
    {
        const int64_t len = lens[(size_t)b];
        if (len - cur_len <= 0)
        {
            rng = cands[(size_t)b].after; // no acceptance draw for a tree that is not longer
            return {true, false};
        }
        Uni g = cands[(size_t)b].after;
        const double draw = g.uni();
        if (accept_worse(energy_step(min_len, cur_len, len), t, log_eps, draw))
        {
            rng = g;
            return {true, true};
        }
        return {false, true};
    }
    int accept(int32_t b)
    {
        std::vector<Edit> e;
        move_edits(topo, moves[(size_t)b], e); // the accepted move's rewrites, on the host either way
        const int rc = commit(e.data(), (int32_t)e.size(), -1);
        cur_len = lens[(size_t)b];
        return rc;
    }
    int32_t batch_limit() const
    {
        const double want = 2.0 * run_len + 2.0;
        return (int32_t)std::max(1.0, std::min((double)max_batch, want));
    }
    void consumed(int32_t k) { run_len = 0.9 * run_len + 0.1 * (double)k; }
};

// StartingTemperature.c:49-195 on the driver
int starting_temperature(Driver &d, double *t_out, int64_t *iterations)
{
    const double log_eps = std::log(LVB_EPS);
    const double increment = 0.00001;
    const int sample = 100;
    double t = LVB_EPS, ratio = 0.0;
    int acc_pos = 0, prop_pos = 0;
    int64_t total = 0;
    while (ratio <= 0.65)
    {
        int iter = 0;
        while (iter <= sample)
        {
            if (iter % REROOT_INTERVAL == 0)
            {
                const int rc = d.reroot();
                if (rc != LVBGPU_OK)
                    return rc;
            }
            // proposals iter .. up to the end of this sample (no re-root tick inside: 101 < 1000)
            d.begin_batch();
            const int32_t lim = std::min<int32_t>(d.batch_limit(), sample + 1 - iter);
            for (int32_t k = 0; k < lim; k++)
                d.speculate(((iter + k) & 1) ? MOVE_SPR : MOVE_NNI, 0.0);
            int rc = d.score();
            if (rc != LVBGPU_OK)
                return rc;
            int32_t used = 0;
            for (int32_t b = 0; b < lim; b++)
            {
                used++;
                const Verdict v = d.decide(b, t, log_eps);
                if (v.worse)
                    prop_pos++;
                if (v.accept)
                {
                    if (v.worse)
                        acc_pos++;
                    rc = d.accept(b);
                    if (rc != LVBGPU_OK)
                        return rc;
                    break;
                }
            }
            d.consumed(used);
            iter += used;
            total += used;
        }
        ratio = (double)acc_pos / prop_pos; // 0/0 gives NaN and ends the loop, as in the reference
        t += increment;
        if (t >= 1 || t <= 0)
        {
            *t_out = 1.0;
            *iterations = total;
            return LVBGPU_OK;
        }
        prop_pos = 0;
        acc_pos = 0;
    }
    *t_out = t - increment;
    *iterations = total;
    return LVBGPU_OK;
}

} // namespace

// ---- C ABI ----------------------------------------------------------------------------------

extern "C" int lvbhost_refrng_new(lvbhost_refrng **out, int32_t seed)
{
    if (!out)
        return LVBGPU_E_ARG;
    lvbhost_refrng *r = new (std::nothrow) lvbhost_refrng();
    if (!r)
        return LVBGPU_E_NOMEM;
    if (!r->g.seed(seed))
    {
        delete r;
        return LVBGPU_E_ARG;
    }
    *out = r;
    return LVBGPU_OK;
}
extern "C" void lvbhost_refrng_free(lvbhost_refrng *r) { delete r; }
extern "C" double lvbhost_refrng_uni(lvbhost_refrng *r) { return r->g.uni(); }
extern "C" int64_t lvbhost_refrng_randpint(lvbhost_refrng *r, int64_t upper) { return r->g.randpint(upper); }

extern "C" int lvbhost_ref_random_tree(lvbhost_refrng *r, int32_t n, int32_t *left, int32_t *right)
{
    if (!r || n < 5 || !left || !right)
        return LVBGPU_E_ARG;
    Topology t;
    std::string why;
    if (!random_tree(n, r->g, t, &why))
        return LVBGPU_E_TOPOLOGY;
    memcpy(left, t.left.data(), (size_t)t.nb * 4);
    memcpy(right, t.right.data(), (size_t)t.nb * 4);
    return LVBGPU_OK;
}

extern "C" int lvbhost_ref_propose(const lvbhost_tree *t, lvbhost_refrng *r, int kind, lvbgpu_edit *edits, int32_t cap,
                                   int32_t *n_edits)
{
    if (!t || !r || !edits || !n_edits || kind < 0 || kind > 2)
        return LVBGPU_E_ARG;
    std::vector<Edit> out;
    std::vector<int32_t> scratch;
    draw_move(t->topo, kind, r->g, out, scratch);
    if ((int32_t)out.size() > cap)
        return LVBGPU_E_ARG;
    memcpy(edits, out.data(), out.size() * sizeof(Edit));
    *n_edits = (int32_t)out.size();
    return LVBGPU_OK;
}

extern "C" int lvbhost_ref_draw_move(const lvbhost_tree *t, lvbhost_refrng *r, int kind, lvbgpu_move *out)
{
    if (!t || !r || !out || kind < 0 || kind > 2)
        return LVBGPU_E_ARG;
    std::vector<int32_t> scratch;
    *out = draw_move_params(t->topo, kind, r->g, scratch);
    return LVBGPU_OK;
}

extern "C" int lvbhost_move_edits(const lvbhost_tree *t, const lvbgpu_move *m, lvbgpu_edit *edits, int32_t cap,
                                  int32_t *n_edits)
{
    if (!t || !m || !edits || !n_edits || m->kind < 0 || m->kind > 2)
        return LVBGPU_E_ARG;
    const Topology &tp = t->topo;
    // the deterministic forms index the arrays with what they are given: check it first
    if (m->kind == MOVE_NNI ? (m->a < tp.n || m->a >= tp.nb) : !spr_move_allowed(tp, m->a, m->b))
        return LVBGPU_E_TOPOLOGY;
    if (m->kind == MOVE_TBR && m->c >= 0)
    {
        std::vector<int32_t> leaves;
        subtree_leaves(tp, m->a, leaves);
        if (std::find(leaves.begin(), leaves.end(), m->c) == leaves.end() || m->c == tp.left[m->a] || m->c == tp.right[m->a])
            return LVBGPU_E_TOPOLOGY;
    }
    std::vector<Edit> out;
    move_edits(tp, *m, out);
    if ((int32_t)out.size() > cap)
        return LVBGPU_E_ARG;
    memcpy(edits, out.data(), out.size() * sizeof(Edit));
    *n_edits = (int32_t)out.size();
    return LVBGPU_OK;
}

extern "C" int lvbhost_ref_arbreroot(const lvbhost_tree *t, lvbhost_refrng *r, lvbgpu_edit *edits, int32_t cap,
                                     int32_t *n_edits, int32_t *new_root)
{
    if (!t || !r || !edits || !n_edits || !new_root)
        return LVBGPU_E_ARG;
    const int32_t nr = draw_new_root(t->topo, r->g);
    std::vector<Edit> out;
    reroot_edits(t->topo, nr, out);
    if ((int32_t)out.size() > cap)
        return LVBGPU_E_ARG;
    memcpy(edits, out.data(), out.size() * sizeof(Edit));
    *n_edits = (int32_t)out.size();
    *new_root = nr;
    return LVBGPU_OK;
}

extern "C" void lvbhost_refsearch_defaults(lvbhost_refsearch_params *p)
{
    if (!p)
        return;
    memset(p, 0, sizeof(*p));
    p->seed = 1;
    p->algorithm = 1;        // SearchParameters.c:81
    p->cooling_schedule = 0; // :77
    p->max_batch = 512;
    p->maxaccept = 5;        // MAXACCEPT_SLOW  LVB.h:148
    p->maxpropose = 2000;    // MAXPROPOSE_SLOW :149
    p->maxfail = 40;         // MAXFAIL_SLOW    :150
}

extern "C" int lvbhost_reference_search(lvbgpu_ctx *ctx, const lvbhost_refsearch_params *p, lvbhost_refsearch_result *res,
                                        lvbhost_tree **tree_out)
{
    if (!ctx || !p || !res || !tree_out)
        return LVBGPU_E_ARG;
    if (p->algorithm < 0 || p->algorithm > 2 || p->cooling_schedule < 0 || p->cooling_schedule > 1 ||
        p->min_len_tree < 0 || p->max_batch < 1)
        return LVBGPU_E_ARG;
    const int32_t n = (int32_t)lvbgpu_n(ctx);
    if (n < 5)
        return LVBGPU_E_ARG;
    memset(res, 0, sizeof(*res));
    const auto wall0 = Clock::now();

    Driver d;
    d.ctx = ctx;
    d.max_batch = std::min<int32_t>(p->max_batch, (int32_t)REROOT_INTERVAL - 1);
    d.min_len = (double)p->min_len_tree;
    static const int default_moves_min = [] {
        const char *e = getenv("LVBHOST_DEVICE_MOVES_MIN");
        return e ? atoi(e) : 128;
    }();
    d.device_moves_min = p->device_moves_min == 0 ? default_moves_min : (int32_t)std::min<int64_t>(p->device_moves_min, 1 << 30);
    d.pb.resize(2 * n - 3);
    if (!d.rng.seed(p->seed))
        return LVBGPU_E_ARG;
    std::string why;

    // GetSoln (Solve.c:536-545): a first random tree only to find the starting temperature ...
    if (!random_tree(n, d.rng, d.topo, &why))
        return LVBGPU_E_TOPOLOGY;
    int rc = d.set_tree();
    if (rc != LVBGPU_OK)
        return rc;
    double t0 = 0.0;
    rc = starting_temperature(d, &t0, &res->st_rearrangements);
    if (rc != LVBGPU_OK)
        return rc;
    res->t0 = t0;
    res->st_scored = d.scored;
    res->st_device_steps = d.steps;

    // ... and a second one to anneal from
    if (!random_tree(n, d.rng, d.topo, &why))
        return LVBGPU_E_TOPOLOGY;
    rc = d.set_tree();
    if (rc != LVBGPU_OK)
        return rc;
    res->start_length = d.cur_len;

    lvbhost_tree *out = new (std::nothrow) lvbhost_tree();
    if (!out)
        return LVBGPU_E_NOMEM;
    BestSet &stack = out->best;
    stack.reset(n);
    std::vector<uint64_t> hscratch;

    LoopState st;
    st.algorithm = p->algorithm;
    st.cooling = p->cooling_schedule;
    st.maxaccept = p->maxaccept;
    st.maxpropose = p->maxpropose;
    st.maxfail = p->maxfail;
    st.start(t0);
    int64_t best = d.cur_len;
    // Anneal() begins by putting the start tree on the stack through the hashing comparison
    // (Solve.c:207, Hash.cpp:49-90): remember its identity for the epilogue below
    std::vector<int32_t> start_identity;
    BestSet::canonical(d.topo, start_identity);
    stack.insert(d.topo);
    int64_t accepted_moves = 0;
    d.run_len = 2.0;

    bool done = false;
    while (!done)
    {
        // re-root tick first (Solve.c:238-247): it is part of the iteration that is about to start
        if ((st.current_iter + 1) % REROOT_INTERVAL == 0)
        {
            rc = d.reroot();
            if (rc != LVBGPU_OK)
                break;
        }
        // speculate: advance a copy of the loop state as if every proposal were rejected
        d.begin_batch();
        LoopState ahead = st;
        const int32_t lim = d.batch_limit();
        for (int32_t k = 0; k < lim; k++)
        {
            if (k > 0 && (ahead.current_iter + 1) % REROOT_INTERVAL == 0)
                break; // the next iteration re-roots first
            ahead.current_iter++;
            const double rv = ahead.draws_for_kind() ? d.rng.uni() : 0.0;
            const int kind = ahead.select(rv);
            d.speculate(kind, rv);
            if (ahead.finish(false))
                break; // frozen after this one if it is rejected
        }
        rc = d.score();
        if (rc != LVBGPU_OK)
            break;
        const int32_t B = (int32_t)d.cands.size();
        int32_t used = 0;
        for (int32_t b = 0; b < B && !done; b++)
        {
            used++;
            st.current_iter++;
            const int kind = st.select(d.cands[(size_t)b].rv);
            if (kind != d.cands[(size_t)b].kind)
            {
                rc = LVBGPU_E_STATE; // speculation and replay disagree: a bug, never a data condition
                done = true;
                break;
            }
            const Verdict v = d.decide(b, st.t, st.log_eps);
            bool new_best_topology = false;
            if (v.accept)
            {
                const int64_t len = d.lens[(size_t)b];
                rc = d.accept(b);
                if (rc != LVBGPU_OK)
                {
                    done = true;
                    break;
                }
                accepted_moves++;
                if (!v.worse && len <= best) // Solve.c:309-320
                {
                    if (len < best)
                        stack.clear();
                    new_best_topology = stack.insert(d.topo);
                }
                if (!v.worse && len < best)
                    best = len;
            }
            if (st.finish(new_best_topology))
                done = true; // frozen (Solve.c:391-401)
            else if (p->max_trees > 0 && (int64_t)stack.kept.size() >= p->max_trees)
                done = true; // Solve.c:447-450
            if (v.accept)
                break; // the rest of the batch were neighbours of the previous tree
        }
        d.consumed(used);
    }
    if (rc != LVBGPU_OK)
    {
        delete out;
        return rc;
    }

    // GetSoln's epilogue (Solve.c:567-569): pop the last tree and offer it back through the
    // hashing comparison, whose memory holds only the start tree's identity
    if (!stack.kept.empty())
    {
        BestSet::Kept last = stack.pop_last();
        if (stack.kept.empty() || last.canon != start_identity)
            stack.push_kept(std::move(last), hscratch);
    }

    out->topo = d.topo;
    out->pb.resize(d.topo.nb);
    out->rng = Rng((uint64_t)p->seed + 1);
    *tree_out = out;
    res->rearrangements = st.current_iter;
    res->best_length = best;
    res->final_length = d.cur_len;
    res->trees = (int64_t)stack.kept.size();
    res->scored = d.scored;
    res->device_steps = d.steps;
    res->accepted_moves = accepted_moves;
    res->reroots = d.reroots;
    res->device_move_steps = d.device_move_steps;
    res->temperatures = st.t_n;
    res->t_final = st.t;
    res->seconds = since(wall0);
    res->seconds_device = d.dev_seconds;
    return LVBGPU_OK;
}
