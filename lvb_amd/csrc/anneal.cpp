// anneal.cpp - batched simulated annealing over the device scoring path.
//
// Mirrors the control flow of the reference's Anneal() (Solve.c:144-479) and
// StartingTemperature() (StartingTemperature.c:49-195): same energy (deltah from
// MinimumTreeLength, Solve.c:303), same acceptance rule (307, 356-377), same counters and
// cooling (380-443), same periodic re-root (240-242).  What differs, by design:
//   * proposals are EDITS scored in a speculative batch: B random neighbours of the current tree
//     are scored in one device step, then consumed in order exactly as the serial loop would
//     consume them; at the first accepted one the tree is committed and the rest of the batch
//     (neighbours of a tree that no longer exists) is discarded.  The chain is distributed as the
//     serial chain; only wasted scoring work differs.  A batch never crosses a temperature
//     change or a re-root tick, and its size follows the running acceptance rate (params.batch is
//     the ceiling).
//   * the re-root is an edit along the old-root..new-root path (the reference re-evaluates the
//     whole tree, TreeOperations.c:631-635); lengths and node sets come out the same.
//   * the treestack finds topologies by a hash of their bipartitions and tells them apart by an exact comparison
//     of canonical forms (host_tree.hpp) instead of sorted object sets; as in the reference (Solve.c:309-320)
//     "accepted" counts proposals that tie or beat the best length AND are a topology not yet in it.
//   * the random stream is xorshift64*, not the reference's Marsaglia generator.
// No length is computed here: all come from lvbgpu_* (HIP).
#include "../../include/lvbhost.h"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "host_tree.hpp"

using namespace lvbgpu;

namespace
{

constexpr double LVB_EPS = 1e-11;     // LVB.h:102
constexpr double FROZEN_T = 0.0001;   // LVB.h:115
constexpr double DBL_EPS = 2.220446049250313e-16;

using Clock = std::chrono::steady_clock;
inline double since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

struct Chain
{
    lvbgpu_ctx *ctx;
    lvbhost_tree *tree;
    std::vector<Edit> edits;
    std::vector<int32_t> offs;
    std::vector<int64_t> lens;
    std::vector<int> kinds;
    double dev_seconds = 0.0;
    int64_t scored = 0, dirty = 0;
    bool on_device = false;          // neighbours drawn on the GPU (lvbgpu_propose_score*)
    std::vector<lvbgpu_edit> fetched; // edits of an accepted device candidate

    std::vector<MoveParams> moves; // host-drawn proposals as parameters
    bool moves_on_device = false;  // ... scored from the parameters (the device built rewrites and programs)
    // below this the host's own program builder (its threads) is quicker than the generator kernel's fixed cost
    static int device_moves_min()
    {
        static const int v = [] {
            const char *e = getenv("LVBHOST_DEVICE_MOVES_MIN");
            return e ? atoi(e) : 96;
        }();
        return v;
    }

    // score B fresh proposals of the given kinds; returns lvbgpu status
    int score(int B)
    {
        static_assert(sizeof(MoveParams) == sizeof(lvbgpu_move), "move layout");
        moves.clear();
        for (int b = 0; b < B; b++)
            moves.push_back(draw_move(tree->topo, kinds[b], tree->rng));
        lens.resize(B);
        moves_on_device = B >= device_moves_min();
        int rc;
        if (moves_on_device)
        {
            const auto t0 = Clock::now();
            rc = lvbgpu_score_moves(ctx, B, reinterpret_cast<const lvbgpu_move *>(moves.data()), lens.data());
            dev_seconds += since(t0);
        }
        else
        {
            edits.clear();
            offs.assign(1, 0);
            for (const MoveParams &m : moves)
            {
                move_edits(tree->topo, m, edits);
                offs.push_back((int32_t)edits.size());
            }
            const auto t0 = Clock::now();
            rc = lvbgpu_score_batch(ctx, B, offs.data(), reinterpret_cast<const lvbgpu_edit *>(edits.data()), nullptr,
                                    lens.data());
            dev_seconds += since(t0);
        }
        scored += B;
        return rc;
    }
    // B neighbours drawn, programmed and scored on the device.  schedule: >= 0 fixed kind,
    // -2 alternate NNI/SPR from `parity`, -3 draw by (p_nni, p_spr)
    int score_device(int B, int schedule, int64_t parity, double p_nni, double p_spr)
    {
        lens.resize(B);
        const uint64_t seed = tree->rng.next();
        const auto t0 = Clock::now();
        int rc;
        if (schedule >= 0)
            rc = lvbgpu_propose_score(ctx, B, schedule, seed, lens.data());
        else
            rc = lvbgpu_propose_score_mixed(ctx, B, p_nni, p_spr, schedule == -2 ? parity : -1, seed, lens.data());
        dev_seconds += since(t0);
        scored += B;
        return rc;
    }
    int commit(int b, int64_t *len)
    {
        if (on_device)
        {
            fetched.resize((size_t)2 * tree->topo.nb + 8);
            int32_t ne = 0;
            const auto t0 = Clock::now();
            int rc = lvbgpu_proposal_edits(ctx, b, fetched.data(), (int32_t)fetched.size(), &ne, nullptr);
            if (rc == LVBGPU_OK)
                rc = lvbgpu_commit(ctx, ne, fetched.data(), -1, nullptr); // asynchronous: length known
            if (rc == LVBGPU_OK)
                *len = lens[b];
            dev_seconds += since(t0);
            if (rc == LVBGPU_OK)
                rc = lvbhost_tree_apply(tree, fetched.data(), ne, -1);
            return rc;
        }
        std::vector<Edit> one;
        if (moves_on_device)
            move_edits(tree->topo, moves[(size_t)b], one); // only the accepted move becomes rewrites on the host
        const Edit *e = moves_on_device ? one.data() : edits.data() + offs[b];
        const int32_t ne = moves_on_device ? (int32_t)one.size() : offs[b + 1] - offs[b];
        const auto t0 = Clock::now();
        // asynchronous: the candidate's length is already known from scoring it
        int rc = lvbgpu_commit(ctx, ne, reinterpret_cast<const lvbgpu_edit *>(e), -1, nullptr);
        if (rc == LVBGPU_OK)
            *len = lens[b];
        dev_seconds += since(t0);
        if (rc == LVBGPU_OK)
            rc = lvbhost_tree_apply(tree, reinterpret_cast<const lvbgpu_edit *>(e), ne, -1);
        return rc;
    }
    int reroot(int64_t *len)
    {
        // arbreroot (TreeOperations.c:639-656): a random leaf other than the current root
        int32_t nr;
        do
            nr = (int32_t)tree->rng.below((uint32_t)tree->topo.n);
        while (nr == tree->topo.root);
        std::vector<Edit> ed;
        reroot_edits(tree->topo, nr, ed);
        const auto t0 = Clock::now();
        int rc = lvbgpu_commit(ctx, (int32_t)ed.size(), reinterpret_cast<const lvbgpu_edit *>(ed.data()), nr, len);
        dev_seconds += since(t0);
        if (rc == LVBGPU_OK)
            rc = lvbhost_tree_apply(tree, reinterpret_cast<const lvbgpu_edit *>(ed.data()), (int32_t)ed.size(), nr);
        return rc;
    }
};

// Metropolis decision of Solve.c:303-378 for a proposal that is worse than the current tree
inline bool accept_worse(double deltah, double t, Rng &rng)
{
    static const double log_eps = std::log(LVB_EPS);
    if (-deltah < t * log_eps)
    {
        (void)rng.uniform(); // the reference draws here too (Solve.c:352-355)
        return false;
    }
    return rng.uniform() < std::exp(-deltah / t);
}

inline double energy_delta(double minlen, int64_t cur, int64_t prop)
{
    double d = minlen / (double)cur - minlen / (double)prop;
    return d > 1.0 ? 1.0 : d;
}

} // namespace

extern "C" void lvbhost_anneal_defaults(lvbhost_anneal_params *p)
{
    if (!p)
        return;
    *p = lvbhost_anneal_params{};
    p->seed = 0x9E3779B97F4A7C15ull;
    p->algorithm = 1;        // SearchParameters.c:81
    p->cooling_schedule = 0; // geometric
    p->batch = 4096; // ceiling: the step size follows the acceptance rate
    p->reroot_interval = 1000;
    p->t0 = 0.0;
    p->maxaccept = 5;
    p->maxpropose = 2000;
    p->maxfail = 40;
    p->min_len_tree = 1;
    p->max_proposals = 0;
    p->max_seconds = 0.0;
    p->max_device_steps = 0;
    p->sync_every = 0;
    p->log_cap = 0;
    p->device_proposals = 2;
    p->run_levels = 0;
    p->lanes = 0;
}

extern "C" int lvbhost_starting_temperature(lvbgpu_ctx *ctx, lvbhost_tree *tree, const lvbhost_anneal_params *p,
                                            double *t0_out)
{
    if (!ctx || !tree || !p || !t0_out)
        return LVBGPU_E_ARG;
    Chain ch{ctx, tree};
    ch.on_device = p->device_proposals == 1;
    int64_t cur = 0;
    int rc = lvbgpu_current_length(ctx, &cur);
    if (rc != LVBGPU_OK)
        return rc;
    const double minlen = (double)p->min_len_tree;
    const int sample = 100; // StartingTemperature.c:86; the loop runs iter = 0..sample inclusive
    const int B = std::max(1, std::min(p->batch, 64));
    double t = LVB_EPS, ratio = 0.0;
    while (ratio <= 0.65)
    {
        int acc_pos = 0, prop_pos = 0;
        int iter = 0;
        while (iter <= sample)
        {
            if (iter % 1000 == 0) // REROOT_INTERVAL: once per temperature (StartingTemperature.c:116-117)
            {
                rc = ch.reroot(&cur);
                if (rc != LVBGPU_OK)
                    return rc;
            }
            const int nb = std::min(B, sample + 1 - iter);
            if (ch.on_device)
                rc = ch.score_device(nb, -2, iter, 0, 0);
            else
            {
                ch.kinds.resize(nb);
                for (int b = 0; b < nb; b++)
                    ch.kinds[b] = ((iter + b) & 1) ? MOVE_SPR : MOVE_NNI; // 123-126
                rc = ch.score(nb);
            }
            if (rc != LVBGPU_OK)
                return rc;
            for (int b = 0; b < nb; b++)
            {
                iter++;
                const int64_t len = ch.lens[b];
                if (len == INT64_MAX)
                    continue; // a device candidate that did not fit its buffers: not a proposal
                bool take = len <= cur;
                if (!take)
                {
                    prop_pos++;
                    take = accept_worse(energy_delta(minlen, cur, len), t, tree->rng);
                    if (take)
                        acc_pos++;
                }
                if (take)
                {
                    rc = ch.commit(b, &cur);
                    if (rc != LVBGPU_OK)
                        return rc;
                    break; // the rest of the batch were neighbours of the old tree
                }
            }
        }
        ratio = (double)acc_pos / prop_pos; // 0/0 gives NaN and ends the loop, as in the reference (StartingTemperature.c:170)
        t += 0.00001; // increment_size
        if (t >= 1 || t <= 0)
        {
            *t0_out = 1.0;
            return LVBGPU_OK;
        }
    }
    *t0_out = t - 0.00001;
    return LVBGPU_OK;
}

extern "C" int lvbhost_anneal(lvbgpu_ctx *ctx, lvbhost_tree *tree, const lvbhost_anneal_params *pp,
                              lvbhost_anneal_result *res, double *log_seconds, int64_t *log_best)
{
    if (!ctx || !tree || !pp || !res)
        return LVBGPU_E_ARG;
    lvbhost_anneal_params p = *pp;
    if (p.batch < 1)
        p.batch = 1;
    const bool lockstep = p.sync_every > 0;
    if (lockstep && p.max_device_steps <= 0)
        return LVBGPU_E_ARG;
    *res = lvbhost_anneal_result{};
    const auto wall0 = Clock::now();
    tree->rng = Rng(p.seed);

    int64_t cur = 0;
    int rc = lvbgpu_current_length(ctx, &cur);
    if (rc != LVBGPU_OK)
        return rc;
    res->start_length = cur;

    double t0 = p.t0;
    if (t0 <= 0.0)
    {
        rc = lvbhost_starting_temperature(ctx, tree, &p, &t0);
        if (rc != LVBGPU_OK)
            return rc;
        rc = lvbgpu_current_length(ctx, &cur);
        if (rc != LVBGPU_OK)
            return rc;
    }

    Chain ch{ctx, tree};
    ch.on_device = p.device_proposals == 1;
    const double minlen = (double)p.min_len_tree;
    const double grad_geom = 0.99, grad_linear = 10 * LVB_EPS; // Solve.c:170-171
    const double log_eps = std::log(LVB_EPS), log_geom = std::log(grad_geom), log_t0 = std::log(t0);
    double t = t0;
    int64_t best = cur, accepted = 0, proposed = 0, failedcnt = 0, t_n = 0;
    int64_t iter = 0;         // Anneal's `iter` (alternation), also our consumed-proposal count
    int64_t current_iter = 0; // *current_iter (re-root ticks)
    double probs[3] = {0, 0, 0}; // trops_probs, Solve.c:210: every proposal is TBR until the first cooling step
    // -a 2 (Solve.c:253-259, 452-466): probabilities follow three counters; in the reference the two
    // kinds NOT tried gain half a count every iteration (its changeAcc flag is only ever set for -a 1)
    double counter[3] = {1, 1, 1};
    auto probs_from_counters = [&] {
        const long total = (long)(counter[0] + counter[1] + counter[2]); // a long there too
        for (int i = 0; i < 3; i++)
            probs[i] = counter[i] / total;
    };
    if (p.algorithm == 2)
        probs_from_counters();
    auto log_point = [&]() {
        if (log_seconds && log_best && res->n_log < p.log_cap)
        {
            log_seconds[res->n_log] = since(wall0);
            log_best[res->n_log] = best;
            res->n_log++;
        }
    };
    log_point();

    tree->best.clear();
    tree->best.insert(tree->topo); // the initial tree is initially the best (Solve.c:208)
    bool done = false;
    double accept_rate = 0.5;
    while (!done)
    {
        // Speculation depth follows the acceptance rate: when most proposals are accepted, all but
        // the first few of a batch would be thrown away; when acceptances are rare the whole
        // batch is consumed.  `accept_rate` is a running estimate per consumed proposal.
        int64_t room = std::min<int64_t>(p.batch, std::max<int64_t>(8, (int64_t)std::ceil(2.0 / accept_rate)));
        // how many proposals may be consumed before something the batch must not straddle
        if (p.reroot_interval > 0)
        {
            const int64_t to_tick = p.reroot_interval - (current_iter % p.reroot_interval);
            // the reference re-roots when the incremented counter hits a multiple (Solve.c:238-242)
            if (to_tick == 1)
            {
                rc = ch.reroot(&cur);
                if (rc != LVBGPU_OK)
                    return rc;
                res->reroots++;
                room = std::min<int64_t>(room, p.reroot_interval);
            }
            else
                room = std::min(room, to_tick - 1);
        }
        room = std::min(room, std::max<int64_t>(1, p.maxpropose - proposed));
        if (p.max_proposals > 0)
            room = std::min(room, std::max<int64_t>(1, p.max_proposals - iter));
        const int B = (int)std::max<int64_t>(1, room);
        if (p.algorithm == 2)
            probs_from_counters(); // per batch here, per iteration in the reference

        // drawing on the device pays once the batch is large (its fixed cost per step is higher,
        // its cost per candidate ~10x lower): mode 2 switches per step
        ch.on_device = p.device_proposals == 1 || (p.device_proposals == 2 && B >= 1024);
        if (ch.on_device)
        {
            switch (p.algorithm)
            {
            case 0: rc = ch.score_device(B, -2, iter, 0, 0); break;
            case 10: rc = ch.score_device(B, MOVE_NNI, 0, 0, 0); break;
            case 11: rc = ch.score_device(B, MOVE_SPR, 0, 0, 0); break;
            case 12: rc = ch.score_device(B, MOVE_TBR, 0, 0, 0); break;
            default: rc = ch.score_device(B, -3, 0, probs[0], probs[1]); break;
            }
            if (rc != LVBGPU_OK)
                return rc;
        }
        ch.kinds.assign(B, -1);
        for (int b = 0; b < B && !ch.on_device; b++)
        {
            int kind;
            switch (p.algorithm)
            {
            case 0: kind = ((iter + b) & 1) ? MOVE_SPR : MOVE_NNI; break; // Solve.c:288-297
            case 10: kind = MOVE_NNI; break;
            case 11: kind = MOVE_SPR; break;
            case 12: kind = MOVE_TBR; break;
            default:
            {
                const double r = tree->rng.uniform(); // Solve.c:262-283
                kind = r < probs[0] ? MOVE_NNI : (r < probs[0] + probs[1] ? MOVE_SPR : MOVE_TBR);
            }
            }
            ch.kinds[b] = kind;
        }
        if (!ch.on_device)
        {
            rc = ch.score(B);
            if (rc != LVBGPU_OK)
                return rc;
        }
        res->device_steps++;

        int consumed_now = 0, accepted_now = 0;
        for (int b = 0; b < B && !done; b++)
        {
            const int64_t len = ch.lens[b];
            if (len == INT64_MAX)
                continue; // a device candidate that did not fit its buffers: not a proposal
            current_iter++;
            consumed_now++;
            if (p.algorithm == 2)
            {
                // host-drawn batches know each proposal's kind; device-drawn ones do not bring it back,
                // there the counters take the expected gain under the batch's probabilities
                const int k = ch.kinds[b];
                for (int i = 0; i < 3; i++)
                    counter[i] += k >= 0 ? (i != k ? 0.5 : 0.0) : 0.5 * (1.0 - probs[i]);
            }
            // accept / reject (Solve.c:303-378)
            bool take;
            if (len <= cur)
                take = true;
            else
                take = accept_worse(energy_delta(minlen, cur, len), t, tree->rng);
            if (take)
            {
                accepted_now++;
                const bool stack_it = len <= cur && len <= best; // ties or beats the best (Solve.c:309)
                rc = ch.commit(b, &cur);
                if (rc != LVBGPU_OK)
                    return rc;
                res->accepted++;
                if (stack_it)
                {
                    if (cur < best)
                        tree->best.clear(); // discard old bests (Solve.c:312-315)
                    if (tree->best.insert(tree->topo))
                        accepted++; // only topologies new to the treestack count (316-319)
                }
                if (cur < best)
                {
                    best = cur;
                    log_point();
                }
            }
            proposed++;
            iter++;

            bool dect = false; // decide whether to reduce temperature (Solve.c:380-407)
            if (accepted >= p.maxaccept)
            {
                failedcnt = 0;
                dect = true;
            }
            else if (proposed >= p.maxpropose)
            {
                failedcnt++;
                if (failedcnt >= p.maxfail && t < FROZEN_T)
                {
                    res->frozen = 1;
                    done = true;
                }
                else
                    dect = true;
            }
            if (dect)
            {
                t_n++;
                if (p.cooling_schedule == 0)
                {
                    const double ln_t = (double)t_n * log_geom + log_t0;
                    t = (ln_t < log_eps) ? LVB_EPS : std::pow(grad_geom, (double)t_n) * t0;
                    if (p.algorithm == 1)
                    {
                        probs[2] = t / t0;
                        probs[1] = (1 - probs[2]) / 2;
                        probs[0] = probs[1];
                    }
                }
                else
                {
                    t = t0 - grad_linear * t_n;
                    if (t < DBL_EPS || t <= LVB_EPS)
                        t = LVB_EPS;
                }
                proposed = 0;
                accepted = 0;
                res->temperatures++;
            }
            if (p.max_proposals > 0 && iter >= p.max_proposals)
                done = true;
            if (take || dect)
                break; // stale neighbours / new temperature: start a fresh batch
        }
        if (consumed_now > 0)
            accept_rate = std::max(1e-4, 0.8 * accept_rate + 0.2 * (double)accepted_now / consumed_now);
        if (p.max_seconds > 0 && since(wall0) >= p.max_seconds)
            done = true;
        if (p.max_device_steps > 0 && res->device_steps >= p.max_device_steps)
            done = true;
        if (lockstep) // every rank runs the same number of batches so the collectives pair up
            done = res->device_steps >= p.max_device_steps;
        if (lockstep && (res->device_steps % p.sync_every == 0 || done))
        {
            int64_t g = best;
            rc = lvbgpu_allreduce_min(ctx, &g, nullptr);
            if (rc != LVBGPU_OK)
                return rc;
            res->global_best_length = g;
        }
    }

    res->best_length = best;
    res->final_length = cur;
    if (p.sync_every <= 0)
        res->global_best_length = best;
    res->scored = ch.scored;
    res->topologies = (int64_t)tree->best.count();
    res->consumed = iter;
    res->t_final = t;
    res->seconds = since(wall0);
    res->seconds_device = ch.dev_seconds;
    return LVBGPU_OK;
}
