// anneal_chains.cpp - R independent annealing chains on ONE GPU, stepped together.
//
// A single chain cannot fill the chip: its step is a chain of small dependent launches (generator, walk, commit) on a
// device that could run a hundred such chains at once, and while most proposals are accepted a step carries only a
// handful of candidates.  The remedy the reference's design offers is the one north_star names for GPUs: independent
// restarts.  Here R of them live in one context (lvbgpu_set_chains: the alignment once, R resident trees) and every
// device step serves all of them: ONE generator launch draws every chain's candidates, ONE walk scores them, every chain's
// first acceptable candidate - the one the serial loop would stop at (Solve.c:300-378) - is picked where the lengths are
// (lvbgpu_chains_step_*: the rule rides with the batch) and ONE commit walk applies every chain's accepted move.  The
// step's latency is paid once, not R times.
//
// Each chain is the loop of anneal.cpp / the reference's Anneal() and StartingTemperature() turned into a state
// machine: plan() says how many candidates of which kinds the chain wants next (speculation depth from its own
// acceptance rate; never across a cooling step or a re-root tick) and by which rule it accepts (current length,
// temperature, a seed for the step's Metropolis draws), consume() books the lengths in order up to the accepted candidate
// the step reports, after_commit() finishes that proposal.  A chain's decisions depend on its own
// state and random stream only, so - for a given run_levels: hot chains draw by the host's law when it is > 0 - its
// trajectory is the same whatever R is (tests/test_gpu_chains.py).
#include "../../include/lvbhost.h"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "decide.h"
#include "host_tree.hpp"

using namespace lvbgpu;

namespace
{

constexpr double LVB_EPS = 1e-11;   // LVB.h:102
constexpr double FROZEN_T = 0.0001; // LVB.h:115
constexpr double DBL_EPS = 2.220446049250313e-16;

using Clock = std::chrono::steady_clock;
inline double since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

inline uint32_t scaled(double p) { return (uint32_t)std::min(4294967295.0, std::max(0.0, p) * 4294967296.0); }

struct ChainRun
{
    int32_t chain = 0;
    lvbgpu_ctx *ctx = nullptr;
    lvbhost_tree *tree = nullptr;
    lvbhost_anneal_params p{};
    lvbhost_anneal_result *res = nullptr;
    double minlen = 1.0;

    enum Phase
    {
        START_TEMP,
        ANNEAL,
        DONE
    } phase = START_TEMP;

    int64_t cur = 0, best = 0;
    // starting temperature (StartingTemperature.c:49-195)
    double st_t = LVB_EPS;
    int st_acc_pos = 0, st_prop_pos = 0, st_iter = 0;
    int64_t st_total = 0; // proposals consumed while the starting temperature was being estimated
    bool st_rerooted = false;
    // annealing (Solve.c:144-479)
    double t0 = 0, t = 0, log_t0 = 0;
    int64_t accepted = 0, proposed = 0, failedcnt = 0, t_n = 0, iter = 0, current_iter = 0;
    double probs[3] = {0, 0, 0}, counter[3] = {1, 1, 1};
    double accept_rate = 0.5;
    double rate_p = 0.5; // acceptances per consumed proposal (a moving average over ~64 proposals): a function of the trajectory alone
    void saw(bool taken) { rate_p += ((taken ? 1.0 : 0.0) - rate_p) * (1.0 / 64.0); }
    // the step in flight
    int B = 0;
    int consumed_now = 0, accepted_now = 0;
    int64_t pending_len = 0; // length of the accepted candidate (after_commit finishes the proposal)
    bool pending_stack = false;
    int rc = LVBGPU_OK;
    std::vector<lvbgpu_edit> fetched;

    void probs_from_counters()
    {
        const long total = (long)(counter[0] + counter[1] + counter[2]); // a long in the reference too
        for (int i = 0; i < 3; i++)
            probs[i] = counter[i] / total;
    }

    // arbreroot (TreeOperations.c:639-656): a random leaf other than the current root.  Only the leaf is drawn here; the
    // driver re-roots all chains that ask for it in one commit walk before it submits the step (lvbgpu_chains_reroot)
    int32_t pending_root = -1;
    void ask_reroot()
    {
        int32_t nr;
        do
            nr = (int32_t)tree->rng.below((uint32_t)tree->topo.n);
        while (nr == tree->topo.root);
        pending_root = nr;
    }
    // the library has re-rooted the chain: follow on the host's own topology
    int rerooted()
    {
        std::vector<Edit> ed;
        reroot_edits(tree->topo, pending_root, ed);
        const int r = lvbhost_tree_apply(tree, reinterpret_cast<const lvbgpu_edit *>(ed.data()), (int32_t)ed.size(), pending_root);
        pending_root = -1;
        res->reroots++;
        return r;
    }

    void begin_anneal(double start_t)
    {
        phase = ANNEAL;
        t0 = t = start_t;
        log_t0 = std::log(t0);
        best = cur;
        if (p.algorithm == 2)
            probs_from_counters();
        tree->best.clear();
        tree->best.insert(tree->topo); // the initial tree is initially the best (Solve.c:208)
    }

    // what the chain wants scored next, and the rule by which it accepts (Solve.c:303-378: lvb_amd/csrc/decide.h); false:
    // nothing (done, or an error in rc)
    static double spec_factor()
    {
        static const double f = [] { const char *e = getenv("LVBHOST_SPEC_FACTOR"); const double v = e ? atof(e) : 4.0; return v > 0.1 ? v : 4.0; }();
        return f;
    }

    bool plan(lvbgpu_chain_draw &d, lvbgpu_chain_rule &rule)
    {
        if (phase == DONE)
            return false;
        d.chain = chain;
        d.mix_a = d.mix_b = 0;
        if (phase == START_TEMP)
        {
            const int sample = 100; // StartingTemperature.c:86; iter runs 0..sample inclusive
            if (st_iter % 1000 == 0 && !st_rerooted) // REROOT_INTERVAL: once per temperature (116-117)
            {
                ask_reroot();
                st_rerooted = true;
            }
            B = std::min(std::max(1, std::min(p.batch, 64)), sample + 1 - st_iter);
            d.kind = -2; // NNI / SPR alternate (123-126)
            d.mix_a = (uint32_t)(st_iter & 1);
        }
        else
        {
            // Speculation depth follows the acceptance rate: when most proposals are accepted, all but the first few
            // of a batch would be thrown away; when acceptances are rare the whole batch is consumed.  4 / rate leaves
            // one step in fifty without an acceptance (e^-4); measured 500 x 50 000, SPR (LVBHOST_SPEC_FACTOR), round 3:
            // factor 2 / 3 / 4 / 6 reach the reference program's 20 s length after 0.270-0.284 / 0.248 / 0.251 / 0.252 s
            // with one chain and 0.466-0.472 / 0.463 / 0.468 / 0.451 s with 32; round 4 (post launch, two lanes), 32 chains,
            // factor 3 / 4 / 6 / 8: 0.399 / 0.386 / 0.381 / 0.402 s, all frozen after 1.209 / 1.211 / 1.248 / 1.326 s
            int64_t room = std::min<int64_t>(p.batch, std::max<int64_t>(8, (int64_t)std::ceil(spec_factor() / accept_rate)));
            if (p.reroot_interval > 0)
            {
                const int64_t to_tick = p.reroot_interval - (current_iter % p.reroot_interval);
                if (to_tick == 1) // the reference re-roots when the incremented counter hits a multiple (Solve.c:238-242)
                {
                    ask_reroot();
                    room = std::min<int64_t>(room, p.reroot_interval);
                }
                else
                    room = std::min(room, to_tick - 1);
            }
            room = std::min(room, std::max<int64_t>(1, p.maxpropose - proposed));
            if (p.max_proposals > 0)
                room = std::min(room, std::max<int64_t>(1, p.max_proposals - iter));
            B = (int)std::max<int64_t>(1, room);
            if (p.algorithm == 2)
                probs_from_counters(); // per batch here, per iteration in the reference
            switch (p.algorithm)
            {
            case 0: // Solve.c:288-297
                d.kind = -2;
                d.mix_a = (uint32_t)(iter & 1);
                break;
            case 10: d.kind = MOVE_NNI; break;
            case 11: d.kind = MOVE_SPR; break;
            case 12: d.kind = MOVE_TBR; break;
            default: // Solve.c:262-283
                d.kind = -3;
                d.mix_a = scaled(probs[0]);
                d.mix_b = scaled(probs[0] + probs[1]);
            }
        }
        d.count = B;
        d.seed = tree->rng.next();
        rule.cur_length = cur;
        rule.temperature = phase == START_TEMP ? st_t : t;
        rule.min_len_tree = minlen;
        rule.accept_seed = tree->rng.next();
        consumed_now = accepted_now = 0;
        res->device_steps++;
        res->scored += B;
        return true;
    }

    // cooling decision after one consumed proposal (Solve.c:380-443); returns true if the temperature changed
    bool after_proposal()
    {
        proposed++;
        iter++;
        bool dect = false;
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
                phase = DONE;
            }
            else
                dect = true;
        }
        if (dect)
        {
            static const double log_eps = std::log(LVB_EPS), grad_geom = 0.99, log_geom = std::log(0.99), grad_linear = 10 * LVB_EPS;
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
            phase = DONE;
        return dect;
    }

    void end_of_start_temperature_sample()
    {
        // (a sample without a single worse proposal: 0/0 = NaN, which ends the reference's loop - StartingTemperature.c:170
        // under `while (r_acc_to_prop <= 0.65)` - so it ends the estimate here too)
        const double ratio = (double)st_acc_pos / st_prop_pos;
        st_t += 0.00001; // increment_size
        st_acc_pos = st_prop_pos = st_iter = 0;
        st_rerooted = false;
        if (st_t >= 1 || st_t <= 0)
            begin_anneal(1.0);
        else if (!(ratio <= 0.65))
            begin_anneal(st_t - 0.00001);
    }

    // Book this step's lengths in order, up to the candidate the step accepted (`pick`, -1: none) - the rest is stale then.
    // The decision itself was made where the lengths are (lvbgpu_chains_step_*); what is counted here is what the serial
    // loop counts (Solve.c:300-378, StartingTemperature.c:128-166).  Returns pick.
    int consume(const int64_t *lens, int32_t pick)
    {
        const int upto = pick >= 0 ? pick + 1 : B;
        if (phase == START_TEMP)
        {
            for (int b = 0; b < upto; b++)
            {
                st_iter++;
                st_total++;
                const int64_t len = lens[b];
                if (len == INT64_MAX)
                    continue; // a device candidate that did not fit its buffers: not a proposal
                if (len > cur)
                {
                    st_prop_pos++;
                    if (b == pick)
                        st_acc_pos++;
                }
                if (b == pick)
                {
                    pending_len = len;
                    return b;
                }
            }
            if (st_iter > 100)
                end_of_start_temperature_sample();
            return -1;
        }
        for (int b = 0; b < upto && phase == ANNEAL; b++)
        {
            const int64_t len = lens[b];
            if (len == INT64_MAX)
                continue;
            current_iter++;
            consumed_now++;
            if (p.algorithm == 2) // the kinds of device-drawn candidates stay on the device: expected gain under the probabilities
                for (int i = 0; i < 3; i++)
                    counter[i] += 0.5 * (1.0 - probs[i]);
            saw(b == pick);
            if (b == pick)
            {
                accepted_now++;
                pending_stack = len <= cur && len <= best; // ties or beats the best (Solve.c:309)
                pending_len = len;
                return b;
            }
            if (after_proposal()) // new temperature: start a fresh batch
            {
                if (pick > b)
                {
                    // (cannot happen: plan() never lets a batch reach past a cooling step, so the temperature can only
                    // change behind the batch's last proposal - the step's rule was the old temperature's)
                    rc = LVBGPU_E_STATE;
                    return -1;
                }
                break;
            }
        }
        end_of_step();
        return -1;
    }

    void end_of_step()
    {
        if (consumed_now > 0)
            accept_rate = std::max(1e-4, 0.8 * accept_rate + 0.2 * (double)accepted_now / consumed_now);
    }

    // The accepted candidate's commit has been ENQUEUED (lvbgpu_chains_commit does not wait): finish the proposal.
    // What has to happen before the next step can be planned needs the new topology in one case only - the accepted tree
    // TIES the best length, so whether it counts as accepted depends on whether the treestack already holds it
    // (Solve.c:316-319).  Everything else (a strictly better tree is new for certain, a worse one is not offered, the
    // starting-temperature phase keeps no trees) is bookkeeping on numbers; the host's copy of the tree and the
    // treestack follow AFTER the next step has been submitted (finish_follow), while the device is busy with it - on the
    // accept path the host is what the device waits for.
    int32_t deferred_pick = -1; // pick index of the move the host's tree has not followed yet
    bool deferred_stack = false;

    // the move's rewrites -> the host's topology (waits for them if they are still on their way from the device)
    void follow(int32_t pick_index)
    {
        fetched.resize((size_t)2 * tree->topo.nb + 8);
        int32_t ne = 0;
        rc = lvbgpu_chains_step_edits(ctx, pick_index, fetched.data(), (int32_t)fetched.size(), &ne);
        if (rc == LVBGPU_OK)
            rc = lvbhost_tree_apply(tree, fetched.data(), ne, -1);
    }
    void finish_follow()
    {
        if (deferred_pick < 0)
            return;
        follow(deferred_pick);
        deferred_pick = -1;
        if (rc == LVBGPU_OK && deferred_stack)
        {
            tree->best.clear(); // discard old bests (Solve.c:312-315)
            tree->best.insert(tree->topo);
        }
        deferred_stack = false;
    }
    // Returns true if the chain's best length improved.
    bool after_commit(int32_t pick_index)
    {
        cur = pending_len;
        if (phase == START_TEMP)
        {
            if (st_iter > 100)
            {
                follow(pick_index); // the estimate may end here, and the annealing starts from the tree as it is
                if (rc == LVBGPU_OK)
                    end_of_start_temperature_sample();
            }
            else
                deferred_pick = pick_index;
            return false;
        }
        res->accepted++;
        bool improved = false;
        if (pending_stack)
        {
            if (cur < best)
            {
                accepted++; // new to the treestack for certain (316-319): the trees themselves follow later
                deferred_pick = pick_index;
                deferred_stack = true;
            }
            else
            {
                follow(pick_index);
                if (rc != LVBGPU_OK)
                    return false;
                if (tree->best.insert(tree->topo))
                    accepted++; // only topologies new to the treestack count
            }
        }
        else
            deferred_pick = pick_index;
        if (cur < best)
        {
            best = cur;
            improved = true;
        }
        (void)after_proposal();
        end_of_step();
        return improved;
    }

    // ------------------------------------------------------------------------------------------------------------
    // RUNS OF ACCEPTANCES IN ONE STEP (one chain per context; lvbhost_anneal_params::run_levels).  While a chain accepts
    // most of what it sees, a device step advances it by ONE move: the candidates behind the accepted one were drawn on
    // the tree before it.  Here the step's candidates are CUMULATIVE: level 1 holds K alternatives drawn on the current
    // tree T (proposals i, i + 1, ..), level 2 K alternatives drawn on T + the first alternative of level 1 (proposals
    // i + 1, i + 2, .. of the run in which proposal i was accepted), and so on: L levels, every candidate given to the
    // scorer as its rewrites relative to T (lvbgpu_score_batch takes any set of rewrites), all scored by one walk.
    // Consumed in order exactly as the serial loop would (Solve.c:300-378): while the FIRST alternative of a level is
    // accepted the next level continues the run; an accepted later alternative, or a level without an acceptance, ends
    // the step.  Proposal i's draw is a function of (seed, i, the tree it is drawn on) and so is its Metropolis draw,
    // so the chain's trajectory does not depend on L or K (tests/test_anneal_chains_cpu.py).  The neighbours are drawn
    // by the host's generators here (proposals.hpp), not by the device's: a chain is in this mode while its acceptance
    // rate is high, and only then.
    double hs_draw_s = 0, hs_consume_s = 0; // LVBHOST_PROFILE: where a host-drawn step's time goes (this chain's share)
    int64_t hs_steps = 0, hs_cands = 0;
    bool hot = false; // in the host-drawn mode (with hysteresis: entered at 0.30 acceptances per proposal, left at 0.15)
    std::vector<Topology> prefix;
    std::vector<Edit> spine, alt, all_edits;
    std::vector<int32_t> offs;
    Topology taken;
    bool wants_host_step(int levels)
    {
        if (levels <= 0 || phase != ANNEAL)
            return hot = false;
        hot = hot ? rate_p >= 0.15 : rate_p >= 0.30;
        return hot;
    }
    static void merge_edits(std::vector<Edit> &into, const Edit *add, size_t n)
    {
        for (size_t i = 0; i < n; i++)
        {
            bool found = false;
            for (Edit &e : into)
                if (e.node == add[i].node)
                {
                    e = add[i];
                    found = true;
                }
            if (!found)
                into.push_back(add[i]);
        }
    }
    int kind_of(uint64_t index, Rng &r, const double *pr) const
    {
        switch (p.algorithm)
        {
        case 0: return (index & 1u) ? MOVE_SPR : MOVE_NNI; // Solve.c:288-297
        case 10: return MOVE_NNI;
        case 11: return MOVE_SPR;
        case 12: return MOVE_TBR;
        default: // Solve.c:262-283
        {
            const double u = r.uniform();
            return u < pr[0] ? MOVE_NNI : (u < pr[0] + pr[1] ? MOVE_SPR : MOVE_TBR);
        }
        }
    }
    // -a 2 (Solve.c:253-259, 452-466): the probabilities follow the move counters, refreshed per proposal in this mode
    // (the device-drawn mode refreshes them per batch); every consumed proposal adds its expected gain, whatever its
    // fate, so the probabilities of proposal i + j are known when proposal i's step is drawn
    static void counters_step(double *cnt, double *pr)
    {
        const long total = (long)(cnt[0] + cnt[1] + cnt[2]);
        for (int i = 0; i < 3; i++)
            pr[i] = cnt[i] / total;
        for (int i = 0; i < 3; i++)
            cnt[i] += 0.5 * (1.0 - pr[i]);
    }
    // A host-drawn step in stages, so that a driver can serve many chains with ONE scoring walk and ONE commit walk:
    //   hs_wants_reroot()  the re-root tick has come: the driver re-roots all such chains in one walk first
    //   hs_draw(levels)    the step's candidates into all_edits / offs (hs_ncand of them); touches this chain only
    //   hs_consume(lens)   books them in order; hs_acc = the candidate the step ends on (-1: none); this chain only
    //   hs_accepted_edits  its rewrites, for the driver's commit; hs_follow() applies them to the host's tree
    int32_t hs_ncand = 0, hs_acc = -1, hs_L = 0, hs_K = 0;
    uint64_t hs_i0 = 0;
    bool hs_wants_reroot()
    {
        if (p.reroot_interval <= 0 || p.reroot_interval - (current_iter % p.reroot_interval) != 1)
            return false;
        ask_reroot(); // the reference re-roots when the incremented counter hits a multiple (Solve.c:238-242)
        return true;
    }
    void hs_draw(int levels, bool rerooted_now)
    {
        auto ht0 = Clock::now();
        int64_t room = INT64_MAX;
        if (p.reroot_interval > 0)
        {
            const int64_t to_tick = rerooted_now ? p.reroot_interval + 1 : p.reroot_interval - (current_iter % p.reroot_interval);
            room = to_tick - 1;
        }
        if (p.max_proposals > 0)
            room = std::min(room, std::max<int64_t>(1, p.max_proposals - iter));
        // alternatives per level: so many that a level goes without an acceptance one time in twenty
        int K = (int)std::ceil(std::log(0.05) / std::log(1.0 - std::min(0.95, std::max(0.05, rate_p))));
        static const int k_max = [] { const char *e = getenv("LVBHOST_RUN_K_MAX"); const int v = e ? atoi(e) : 8; return v < 1 ? 1 : (v > 8 ? 8 : v); }();
        K = std::max(std::min(2, k_max), std::min(K, k_max));
        int L = std::max(1, std::min(levels, 8));
        while (L > 1 && (int64_t)(L - 1) + K > room)
            L--;
        K = (int)std::max<int64_t>(1, std::min<int64_t>(K, room - (L - 1)));
        // the candidates
        prefix.resize((size_t)L);
        prefix[0] = tree->topo;
        spine.clear();
        all_edits.clear();
        offs.assign(1, 0);
        const uint64_t i0 = (uint64_t)iter;
        std::string why;
        // the move probabilities of proposals i0, i0 + 1, ..: constant within a step (a cooling step ends it, below)
        // except with -a 2, where they follow the counters proposal by proposal
        double pr_at[16][3];
        {
            double cnt[3] = {counter[0], counter[1], counter[2]}, pr[3] = {probs[0], probs[1], probs[2]};
            for (int o = 0; o < L - 1 + K && o < 16; o++)
            {
                if (p.algorithm == 2)
                    counters_step(cnt, pr);
                for (int i = 0; i < 3; i++)
                    pr_at[o][i] = pr[i];
            }
        }
        for (int d = 0; d < L; d++)
        {
            const size_t first_off = all_edits.size();
            size_t first_n = 0;
            for (int k = 0; k < K; k++)
            {
                const uint64_t gi = i0 + (uint64_t)d + (uint64_t)k;
                Rng r(lvb_mix64(p.seed * 0x9E3779B97F4A7C15ull + gi + 1u));
                const int kind = kind_of(gi, r, pr_at[d + k]);
                alt.clear();
                propose(prefix[(size_t)d], kind, r, alt);
                const size_t at = all_edits.size();
                all_edits.insert(all_edits.end(), spine.begin(), spine.end());
                {
                    std::vector<Edit> cum(all_edits.begin() + (long)at, all_edits.end());
                    merge_edits(cum, alt.data(), alt.size());
                    all_edits.resize(at);
                    all_edits.insert(all_edits.end(), cum.begin(), cum.end());
                }
                offs.push_back((int32_t)all_edits.size());
                if (k == 0)
                    first_n = all_edits.size() - at;
            }
            if (d + 1 < L) // the run goes on from the first alternative
            {
                spine.assign(all_edits.begin() + (long)first_off, all_edits.begin() + (long)(first_off + first_n));
                prefix[(size_t)d + 1] = tree->topo;
                if (!tree->pb.apply_edits(prefix[(size_t)d + 1], spine.data(), (int32_t)spine.size(), -1, &why))
                {
                    rc = LVBGPU_E_TOPOLOGY;
                    return;
                }
            }
        }
        hs_L = L;
        hs_K = K;
        hs_i0 = i0;
        hs_ncand = L * K;
        hs_acc = -1;
        hs_draw_s += since(ht0);
        hs_steps++;
        hs_cands += hs_ncand;
    }
    void hs_consume(const int64_t *lens)
    {
        auto ht0 = Clock::now();
        const int L = hs_L, K = hs_K;
        const uint64_t i0 = hs_i0;
        res->device_steps++;
        res->host_steps++;
        res->scored += hs_ncand;
        consumed_now = accepted_now = 0;
        int acc = -1; // the candidate the step ends on (its rewrites are the step's commit)
        DecideRule rule{};
        rule.minlen = minlen;
        rule.seed = lvb_mix64(p.seed ^ 0xACCE97EDull);
        std::string why;
        bool cooled = false;
        for (int d = 0; d < L && phase == ANNEAL; d++)
        {
            bool run_goes_on = false;
            for (int k = 0; k < K && phase == ANNEAL; k++)
            {
                const int c = d * K + k;
                const int64_t len = lens[c];
                current_iter++;
                consumed_now++;
                if (p.algorithm == 2)
                    counters_step(counter, probs);
                rule.cur = cur;
                rule.t = t;
                const bool take = lvb_take((long long)len, &rule, (uint32_t)(i0 + (uint64_t)d + (uint64_t)k)) != 0;
                saw(take);
                if (rate_p < 0.15) // the chain has cooled off: device-drawn steps from the next proposal on - at the same
                    cooled = true; // proposal whatever the step's shape
                if (!take)
                {
                    if (after_proposal()) // a cooling step: what follows was drawn under the old temperature's probabilities
                        cooled = true;
                    if (cooled)
                        break;
                    continue;
                }
                accepted_now++;
                res->accepted++;
                if (len <= cur && len <= best) // ties or beats the best (Solve.c:309-319): the treestack needs the tree
                {
                    taken = tree->topo;
                    if (!tree->pb.apply_edits(taken, all_edits.data() + offs[(size_t)c], offs[(size_t)c + 1] - offs[(size_t)c], -1, &why))
                    {
                        rc = LVBGPU_E_TOPOLOGY;
                        return;
                    }
                    if (len < best)
                        tree->best.clear();
                    if (tree->best.insert(taken))
                        accepted++;
                }
                cur = len;
                if (cur < best)
                    best = cur;
                acc = c;
                if (after_proposal())
                    cooled = true;
                run_goes_on = k == 0 && !cooled;
                break;
            }
            if (!run_goes_on || cooled)
                break;
        }
        accept_rate = std::max(1e-4, rate_p); // (what the device-drawn steps size their batches by: not a function of this step's shape)
        hs_acc = acc;
        hs_consume_s += since(ht0);
    }
    // the host's tree takes the accepted candidate's rewrites (the driver has enqueued their commit)
    void hs_follow()
    {
        if (hs_acc < 0 || rc != LVBGPU_OK)
            return;
        rc = lvbhost_tree_apply(tree, reinterpret_cast<const lvbgpu_edit *>(all_edits.data()) + offs[(size_t)hs_acc],
                                offs[(size_t)hs_acc + 1] - offs[(size_t)hs_acc], -1);
    }

    void finish()
    {
        res->best_length = best;
        res->final_length = cur;
        res->global_best_length = best;
        res->topologies = (int64_t)tree->best.count();
        res->consumed = iter;
        res->t_final = t;
    }
};

} // namespace

namespace
{
// One LANE: some of the run's chains in a context of their own (own stream, own batches), stepped together.  The lanes'
// steps overlap on the device: while one lane's lengths are on their way to the host, are consumed and its next step is
// planned and submitted, the device walks the other lane's candidates.
struct Lane
{
    lvbgpu_ctx *ctx = nullptr;
    bool forked = false;       // made here (lvbgpu_fork), destroyed here
    int32_t first = 0, count = 0; // chains [first, first + count) of the run, chain c being slot c - first of ctx
    // the step in flight
    std::vector<lvbgpu_chain_draw> draws;
    std::vector<lvbgpu_chain_rule> rules;
    std::vector<int32_t> who; // global chain numbers, by draw
    std::vector<int64_t> lens;
    std::vector<int32_t> picks;
    size_t total = 0;
    bool active = false;
    // the two batch slots take turns: the step being submitted draws into one while the chains' accepted candidates of
    // the step before still lie in the other - their commit walk rides in the SAME launch as this step's generator
    // (the library's post launch), which it could not if the generator overwrote their programs
    int32_t slot = 0;
    bool host_stepped = false;
    // the host part of a step (runs of acceptances while chains are hot)
    std::vector<ChainRun *> hot;
    std::vector<uint8_t> hot_rerooted;
    std::vector<int32_t> h_chain_of, h_offs, h_first, h_commit_chain, h_commit_offs;
    std::vector<lvbgpu_edit> h_edits, h_commit_edits;
    std::vector<int64_t> h_lens;
    std::vector<lvbgpu_chain_root> roots;
    int64_t steps = 0;
};
struct LaneGuard // the forked contexts go with the run, whichever way it ends
{
    std::vector<Lane> &lanes;
    ~LaneGuard()
    {
        for (Lane &l : lanes)
            if (l.forked && l.ctx)
                lvbgpu_destroy(l.ctx);
    }
};

// the chains of one run, in one or more lanes; wall0: the run's clock origin
int anneal_chains_run(lvbgpu_ctx *ctx, int32_t R, lvbhost_tree *const *trees, const lvbhost_anneal_params *params,
                      lvbhost_anneal_result *results, double *log_seconds, int64_t *log_best, int32_t *n_log,
                      const Clock::time_point wall0)
{
    if (!ctx || R < 1 || R > 64 || !trees || !params || !results)
        return LVBGPU_E_ARG;
    for (int32_t c = 0; c < R; c++)
        if (!trees[c])
            return LVBGPU_E_ARG;
    const bool lockstep = params[0].sync_every > 0;
    if (lockstep && params[0].max_device_steps <= 0)
        return LVBGPU_E_ARG;
    // How many lanes.  A step is a chain: scoring walk -> lengths to the host -> the host consumes, plans, submits ->
    // post launch (commit walk, table rebuilds, generator) -> scoring walk.  With all chains in one lane the device idles
    // while the host works and only the post launch's few workgroups run between two walks: at 32 chains of 500 x 50 000
    // the walk is 62 % of a step (rocprofv3: walk 102 us, post launch 35 us, host gap 27 us).  Two lanes take turns: one
    // lane's walk covers the other's host work and post launch.  (Round 2/3 measured two groups taking turns on ONE
    // stream - a loss, everything of one group waits behind the other's walk - and groups with host threads and contexts
    // of their own - no gain; here ONE host thread serves whichever lane's lengths have arrived.)  From 16 chains on by
    // default (a lane of few chains does not fill the chip); lockstep runs keep one lane (their step count is the
    // collectives' clock).
    int32_t nl = params[0].lanes > 0 ? params[0].lanes : (R >= 16 ? 2 : 1);
    if (const char *e = getenv("LVBHOST_LANES"))
        if (atoi(e) > 0)
            nl = atoi(e);
    nl = std::max(1, std::min({nl, R, 8}));
    if (lockstep)
        nl = 1;
    std::vector<Lane> lanes((size_t)nl);
    LaneGuard guard{lanes};
    int rc = LVBGPU_OK;
    for (int32_t g = 0; g < nl && rc == LVBGPU_OK; g++)
    {
        Lane &L = lanes[(size_t)g];
        L.first = (int32_t)((int64_t)R * g / nl);
        L.count = (int32_t)((int64_t)R * (g + 1) / nl) - L.first;
        if (g == 0)
            L.ctx = ctx;
        else
        {
            rc = lvbgpu_fork(ctx, &L.ctx);
            L.forked = rc == LVBGPU_OK;
        }
        if (rc == LVBGPU_OK && lvbgpu_chains(L.ctx) != L.count)
            rc = lvbgpu_set_chains(L.ctx, L.count);
    }
    if (rc != LVBGPU_OK)
        return rc;
    std::vector<ChainRun> runs((size_t)R);
    std::vector<int32_t> lane_of((size_t)R, 0);
    for (int32_t g = 0; g < nl; g++)
        for (int32_t c = lanes[(size_t)g].first; c < lanes[(size_t)g].first + lanes[(size_t)g].count; c++)
            lane_of[(size_t)c] = g;
    for (int32_t c = 0; c < R && rc == LVBGPU_OK; c++)
    {
        Lane &L = lanes[(size_t)lane_of[(size_t)c]];
        ChainRun &r = runs[(size_t)c];
        r.chain = c - L.first;
        r.ctx = L.ctx;
        r.tree = trees[c];
        r.p = params[c];
        if (r.p.batch < 1)
            r.p.batch = 1;
        r.res = &results[c];
        *r.res = lvbhost_anneal_result{};
        r.minlen = (double)r.p.min_len_tree;
        r.tree->rng = Rng(r.p.seed);
        rc = lvbgpu_select_chain(L.ctx, r.chain);
        if (rc == LVBGPU_OK)
            rc = lvbhost_tree_upload(L.ctx, r.tree, &r.cur);
        r.res->start_length = r.cur;
        if (r.p.t0 > 0.0)
            r.begin_anneal(r.p.t0);
    }
    if (rc != LVBGPU_OK)
        return rc;
    const int32_t log_cap = params[0].log_cap;
    int32_t nlog = 0;
    int64_t global_best = INT64_MAX;
    auto log_point = [&] {
        // best length of the annealing proper: what a chain passes through while its starting temperature is being
        // estimated does not count (the reference's best starts with Anneal(), Solve.c:208)
        int64_t g = INT64_MAX;
        for (const ChainRun &r : runs)
            if (r.phase != ChainRun::START_TEMP)
                g = std::min(g, r.best);
        if (g < global_best)
        {
            global_best = g;
            if (log_seconds && log_best && log_cap > 0)
            {
                if (nlog == log_cap)
                    nlog--; // a full log keeps its last entry current: the log always ends at the best length reached
                log_seconds[nlog] = since(wall0);
                log_best[nlog] = g;
                nlog++;
            }
        }
    };
    log_point();

    int64_t steps = 0, busy_scored = 0;
    double dev_seconds = 0.0, busy_seconds = 0.0;
    double t_plan = 0, t_score = 0, t_consume = 0, t_after = 0; // LVBHOST_PROFILE=1 prints them
    double t_submit = 0, t_reroot = 0, t_idle = 0; // ... and, of those, the submit call (part of propose_score), the re-roots (part of plan); waiting for any lane
    const bool profile = getenv("LVBHOST_PROFILE") != nullptr;
    int64_t p_sc = 0, p_co = 0, p_ac = 0, p_cs = 0;
    auto chain_steps_now = [&] {
        int64_t v = 0;
        for (const ChainRun &r : runs)
            v += r.res->device_steps;
        return v;
    };

    // Runs of acceptances in one step while chains are hot (ChainRun::hs_*): the hot chains of a lane's step are served by
    // the host part below - their cumulative candidates drawn here (on the context's host threads), ONE scoring walk for
    // all of them (lvbgpu_chains_score_edits), ONE commit walk for what they accept (lvbgpu_chains_commit_edits) - the
    // others by the device step as before; a step may have both parts.
    const int levels = std::max(0, (int)params[0].run_levels);
    double hs_score_s = 0, hs_commit_s = 0, hs_prep_s = 0;
    int64_t hs_parts = 0;
    struct HotJob
    {
        std::vector<ChainRun *> *hot;
        std::vector<uint8_t> *rerooted;
        const int64_t *lens;
        const int32_t *first;
        int levels;
    };
    // the host part of a step: everything for the lane's hot chains, start to finish (nothing of it stays in flight)
    auto host_part = [&](Lane &L) -> int {
        HotJob hot_job{&L.hot, &L.hot_rerooted, nullptr, nullptr, levels};
        auto t0 = Clock::now();
        // re-roots first (the candidates are drawn on the re-rooted trees), all of them in one walk
        L.roots.clear();
        L.hot_rerooted.assign(L.hot.size(), 0);
        for (size_t i = 0; i < L.hot.size(); i++)
        {
            L.hot[i]->finish_follow();
            if (L.hot[i]->rc != LVBGPU_OK)
                return L.hot[i]->rc;
            if (L.hot[i]->hs_wants_reroot())
            {
                L.roots.push_back({L.hot[i]->chain, L.hot[i]->pending_root});
                L.hot_rerooted[i] = 1;
            }
        }
        if (!L.roots.empty())
        {
            int rr = lvbgpu_chains_reroot(L.ctx, (int32_t)L.roots.size(), L.roots.data());
            for (size_t i = 0; i < L.hot.size() && rr == LVBGPU_OK; i++)
                if (L.hot_rerooted[i])
                    rr = L.hot[i]->rerooted();
            if (rr != LVBGPU_OK)
                return rr;
        }
        int rp = lvbgpu_parallel_for(L.ctx, (int32_t)L.hot.size(), [](int32_t i, void *a) {
            HotJob *j = (HotJob *)a;
            (*j->hot)[(size_t)i]->hs_draw(j->levels, (*j->rerooted)[(size_t)i] != 0); }, &hot_job);
        if (rp != LVBGPU_OK)
            return rp;
        L.h_chain_of.clear();
        L.h_offs.assign(1, 0);
        L.h_edits.clear();
        L.h_first.clear();
        for (ChainRun *r : L.hot)
        {
            if (r->rc != LVBGPU_OK)
                return r->rc;
            L.h_first.push_back((int32_t)L.h_chain_of.size());
            const int32_t base = (int32_t)L.h_edits.size();
            for (int32_t c = 0; c < r->hs_ncand; c++)
            {
                L.h_chain_of.push_back(r->chain);
                L.h_offs.push_back(base + r->offs[(size_t)c + 1]);
            }
            const lvbgpu_edit *e = reinterpret_cast<const lvbgpu_edit *>(r->all_edits.data());
            L.h_edits.insert(L.h_edits.end(), e, e + r->all_edits.size());
        }
        L.h_lens.resize(L.h_chain_of.size());
        hs_prep_s += since(t0);
        t0 = Clock::now();
        int rs = lvbgpu_chains_score_edits(L.ctx, (int32_t)L.h_chain_of.size(), L.h_chain_of.data(), L.h_offs.data(), L.h_edits.data(), L.h_lens.data());
        if (rs != LVBGPU_OK)
            return rs;
        hs_score_s += since(t0);
        t0 = Clock::now();
        hot_job.lens = L.h_lens.data();
        hot_job.first = L.h_first.data();
        rp = lvbgpu_parallel_for(L.ctx, (int32_t)L.hot.size(), [](int32_t i, void *a) {
            HotJob *j = (HotJob *)a;
            (*j->hot)[(size_t)i]->hs_consume(j->lens + j->first[(size_t)i]); }, &hot_job);
        if (rp != LVBGPU_OK)
            return rp;
        L.h_commit_chain.clear();
        L.h_commit_offs.assign(1, 0);
        L.h_commit_edits.clear();
        for (ChainRun *r : L.hot)
        {
            if (r->rc != LVBGPU_OK)
                return r->rc;
            if (r->hs_acc < 0)
                continue;
            const lvbgpu_edit *e = reinterpret_cast<const lvbgpu_edit *>(r->all_edits.data());
            L.h_commit_chain.push_back(r->chain);
            L.h_commit_edits.insert(L.h_commit_edits.end(), e + r->offs[(size_t)r->hs_acc], e + r->offs[(size_t)r->hs_acc + 1]);
            L.h_commit_offs.push_back((int32_t)L.h_commit_edits.size());
        }
        if (!L.h_commit_chain.empty())
        {
            const int rcm = lvbgpu_chains_commit_edits(L.ctx, (int32_t)L.h_commit_chain.size(), L.h_commit_chain.data(), L.h_commit_offs.data(),
                                                       L.h_commit_edits.data());
            if (rcm != LVBGPU_OK)
                return rcm;
            rp = lvbgpu_parallel_for(L.ctx, (int32_t)L.hot.size(), [](int32_t i, void *a) { (*((HotJob *)a)->hot)[(size_t)i]->hs_follow(); }, &hot_job);
            if (rp != LVBGPU_OK)
                return rp;
            for (ChainRun *r : L.hot)
                if (r->rc != LVBGPU_OK)
                    return r->rc;
        }
        hs_commit_s += since(t0);
        hs_parts++;
        return LVBGPU_OK;
    };

    // plan the lane's chains and enqueue their step (nothing if every chain is done)
    auto submit_step = [&](Lane &L) -> int {
        L.hot.clear();
        L.host_stepped = false;
        if (levels > 0)
            for (int32_t c = L.first; c < L.first + L.count; c++)
                if (runs[(size_t)c].wants_host_step(levels))
                    L.hot.push_back(&runs[(size_t)c]);
        L.draws.clear();
        L.rules.clear();
        L.who.clear();
        L.total = 0;
        L.active = false;
        auto tp = Clock::now();
        for (int32_t c = L.first; c < L.first + L.count; c++)
        {
            ChainRun &r = runs[(size_t)c];
            lvbgpu_chain_draw d{};
            lvbgpu_chain_rule rule{};
            if (r.hot)
                continue; // (served by the host part)
            if (r.plan(d, rule))
            {
                L.draws.push_back(d);
                L.rules.push_back(rule);
                L.who.push_back(c);
                L.total += (size_t)d.count;
            }
            else if (r.rc != LVBGPU_OK)
                return r.rc;
        }
        // the chains whose re-root tick has come: all of them in one commit walk, before their candidates are drawn
        L.roots.clear();
        for (int32_t c : L.who)
            if (runs[(size_t)c].pending_root >= 0)
            {
                runs[(size_t)c].finish_follow(); // the re-root's rewrites are made from the tree as it is NOW
                if (runs[(size_t)c].rc != LVBGPU_OK)
                    return runs[(size_t)c].rc;
                L.roots.push_back({runs[(size_t)c].chain, runs[(size_t)c].pending_root});
            }
        if (!L.roots.empty())
        {
            auto tr = Clock::now();
            int rr = lvbgpu_chains_reroot(L.ctx, (int32_t)L.roots.size(), L.roots.data());
            t_reroot += since(tr);
            for (size_t i = 0; i < L.roots.size() && rr == LVBGPU_OK; i++)
                rr = runs[(size_t)(L.first + L.roots[i].chain)].rerooted();
            if (rr != LVBGPU_OK)
                return rr;
        }
        t_plan += since(tp);
        int r = LVBGPU_OK;
        if (!L.draws.empty())
        {
            L.lens.resize(L.total);
            L.picks.resize(L.draws.size());
            auto td = Clock::now();
            L.slot ^= 1;
            r = lvbgpu_chains_step_submit(L.ctx, L.slot, (int32_t)L.draws.size(), L.draws.data(), L.rules.data());
            dev_seconds += since(td);
            t_score += since(td);
            t_submit += since(td);
            L.active = r == LVBGPU_OK;
        }
        if (r == LVBGPU_OK && !L.hot.empty()) // (while the device draws and walks the others' candidates)
        {
            auto td = Clock::now();
            r = host_part(L);
            dev_seconds += since(td);
            t_score += since(td);
            L.host_stepped = r == LVBGPU_OK;
        }
        return r;
    };
    // the trees of the lane's chains that accepted in the step before follow their moves now, while the device works
    auto follow_all = [&](Lane &L) -> int {
        auto tf = Clock::now();
        for (int32_t c = L.first; c < L.first + L.count; c++)
        {
            ChainRun &r = runs[(size_t)c];
            if (r.deferred_pick >= 0)
            {
                r.finish_follow();
                if (r.rc != LVBGPU_OK)
                    return r.rc;
            }
        }
        t_after += since(tf);
        return LVBGPU_OK;
    };
    // after every step: the log, when each chain stopped, until when at least half of them were still at it
    auto account = [&] {
        log_point(); // R comparisons: nothing next to a device step
        int32_t active = 0;
        int64_t scored_now = 0;
        const double now = since(wall0);
        for (ChainRun &r : runs)
        {
            scored_now += r.res->scored;
            if (r.phase != ChainRun::DONE)
                active++;
            else if (r.res->seconds_done == 0.0)
                r.res->seconds_done = now;
        }
        if (2 * active >= R)
        {
            busy_seconds = now;
            busy_scored = scored_now;
        }
    };
    // the lane's lengths and picks are back (or are waited for): every chain books its own; the accepted moves are
    // committed already (lvbgpu_chains_step_*)
    auto finish_step = [&](Lane &L) -> int {
        if (!L.active)
            return LVBGPU_OK;
        L.active = false;
        auto td = Clock::now();
        int r = lvbgpu_chains_step_collect(L.ctx, L.slot, L.lens.data(), L.picks.data());
        dev_seconds += since(td);
        t_score += since(td);
        if (r != LVBGPU_OK)
            return r;
        size_t off = 0;
        auto tp = Clock::now();
        for (size_t i = 0; i < L.draws.size(); i++)
        {
            ChainRun &cr = runs[(size_t)L.who[i]];
            const int b = cr.consume(L.lens.data() + off, L.picks[i]);
            off += (size_t)L.draws[i].count;
            if (cr.rc != LVBGPU_OK)
                return cr.rc;
            if (b >= 0)
            {
                (void)cr.after_commit((int32_t)i);
                if (cr.rc != LVBGPU_OK)
                    return cr.rc;
            }
        }
        t_consume += since(tp);
        return LVBGPU_OK;
    };
    // which lane next: one whose step is the host's alone, else the first whose lengths have arrived (asked in turn,
    // starting behind the lane served last; never waits inside the library - memory is polled)
    int32_t last_served = nl - 1;
    auto next_lane = [&](int32_t *out) -> int {
        int32_t live = 0;
        for (const Lane &L : lanes)
            live += (L.active || L.host_stepped) ? 1 : 0;
        if (!live)
        {
            *out = -1;
            return LVBGPU_OK;
        }
        auto tw = Clock::now();
        for (uint64_t spins = 1;; spins++)
        {
            for (int32_t k = 1; k <= nl; k++)
            {
                const int32_t g = (last_served + k) % nl;
                Lane &L = lanes[(size_t)g];
                int32_t ready = 0;
                if (!L.active)
                    ready = L.host_stepped ? 1 : 0;
                else if (nl == 1)
                    ready = 1; // (the collect waits, with the library's own deadline)
                else
                {
                    const int rr = lvbgpu_chains_ready(L.ctx, L.slot, &ready);
                    if (rr != LVBGPU_OK)
                        return rr;
                }
                if (ready)
                {
                    *out = g;
                    t_idle += since(tw);
                    return LVBGPU_OK;
                }
            }
            if ((spins & 0xFFFFu) == 0 && since(tw) > 60.0) // (the collect reports what is stuck)
            {
                for (int32_t g = 0; g < nl; g++)
                    if (lanes[(size_t)g].active)
                    {
                        *out = g;
                        return LVBGPU_OK;
                    }
            }
        }
    };

    for (Lane &L : lanes)
    {
        rc = submit_step(L);
        if (rc != LVBGPU_OK)
            break;
    }
    while (rc == LVBGPU_OK)
    {
        int32_t g = -1;
        rc = next_lane(&g);
        if (rc != LVBGPU_OK)
            break;
        if (g < 0)
        {
            if (!lockstep)
                break;
            g = 0; // a lockstep run keeps stepping (nothing to score) so that the collectives pair up
        }
        Lane &L = lanes[(size_t)g];
        last_served = g;
        if (L.active)
            rc = finish_step(L);
        L.host_stepped = false;
        if (rc != LVBGPU_OK)
            break;
        account();
        L.steps++;
        steps++;
        if (profile && steps % (1000 * (int64_t)nl) == 0) // how full the batches are and how often a proposal is accepted, as the run goes
        {
            int64_t sc = 0, co = 0, ac = 0;
            for (const ChainRun &r : runs)
            {
                sc += r.res->scored;
                co += r.iter + r.st_total;
                ac += r.res->accepted;
            }
            fprintf(stderr, "[anneal_chains] step %lld at %.3f s: scored %lld (+%lld), consumed +%lld, accepted +%lld  => %.1f candidates per chain-step, "
                            "acceptance %.3f per consumed proposal\n",
                    (long long)(steps / nl), since(wall0), (long long)sc, (long long)(sc - p_sc), (long long)(co - p_co), (long long)(ac - p_ac),
                    (double)(sc - p_sc) / std::max<int64_t>(1, chain_steps_now() - p_cs), (double)(ac - p_ac) / std::max<int64_t>(1, co - p_co));
            p_sc = sc;
            p_co = co;
            p_ac = ac;
            p_cs = chain_steps_now();
        }
        bool stop = false;
        if (params[0].max_seconds > 0 && since(wall0) >= params[0].max_seconds)
            stop = true;
        if (params[0].max_device_steps > 0 && L.steps >= params[0].max_device_steps)
            stop = true; // (every lane gets that many steps: the others run on until they have theirs)
        if (lockstep) // every rank runs the same number of steps so that the collectives pair up (one lane)
        {
            stop = steps >= params[0].max_device_steps;
            if (steps % params[0].sync_every == 0 || stop)
            {
                int64_t gb = global_best; // INT64_MAX while every chain is still estimating its starting temperature: "no length"
                rc = lvbgpu_allreduce_min(ctx, &gb, nullptr);
                if (rc != LVBGPU_OK)
                    break;
                for (ChainRun &r : runs)
                    r.res->global_best_length = gb;
            }
        }
        if (stop)
        {
            if (params[0].max_seconds > 0 && since(wall0) >= params[0].max_seconds)
                break; // out of time: whatever the other lanes have in flight is dropped with the run
            if (lockstep)
                break;
            continue; // this lane has had its steps; the others finish theirs
        }
        rc = submit_step(L);
        if (rc == LVBGPU_OK)
            rc = follow_all(L);
    }
    // what is still in flight (a run that ran out of time while other lanes were at work) is booked like any other step -
    // the collect commits its picks, so the chains' counters have to follow - and the trees follow their last moves
    for (Lane &L : lanes)
    {
        if (rc == LVBGPU_OK && L.active)
        {
            rc = finish_step(L);
            if (rc == LVBGPU_OK)
            {
                account();
                L.steps++;
                steps++;
            }
        }
        if (rc == LVBGPU_OK)
            rc = follow_all(L); // (a run that stopped right after a step)
    }
    if (rc != LVBGPU_OK)
        return rc;
    const double secs = since(wall0);
    if (profile && steps > 0)
        fprintf(stderr, "[anneal_chains] R=%d lanes=%d lane-steps=%lld  per lane-step (us): plan %.1f (re-roots %.1f)  step submit + collect %.1f (submit %.1f)  "
                        "consume + bookkeeping %.1f  trees following %.1f  waiting for a lane %.1f  wall per round of all lanes %.1f\n",
                R, nl, (long long)steps, 1e6 * t_plan / steps, 1e6 * t_reroot / steps, 1e6 * t_score / steps, 1e6 * t_submit / steps,
                1e6 * t_consume / steps, 1e6 * t_after / steps, 1e6 * t_idle / steps, 1e6 * secs * nl / steps);
    if (profile && hs_parts)
    {
        int64_t chain_steps = 0, cands = 0;
        double draw_s = 0, consume_s = 0;
        for (const ChainRun &r : runs)
        {
            chain_steps += r.hs_steps;
            cands += r.hs_cands;
            draw_s += r.hs_draw_s;
            consume_s += r.hs_consume_s;
        }
        fprintf(stderr, "[anneal_chains] host-drawn parts: %lld (%.1f chains, %.1f candidates per chain); per part (us): re-roots + draw + gather %.1f  "
                        "score %.1f  consume + commit %.1f   (summed over the chains' threads: draw %.1f, consume %.1f)\n",
                (long long)hs_parts, (double)chain_steps / hs_parts, (double)cands / std::max<int64_t>(1, chain_steps), 1e6 * hs_prep_s / hs_parts,
                1e6 * hs_score_s / hs_parts, 1e6 * hs_commit_s / hs_parts, 1e6 * draw_s / hs_parts, 1e6 * consume_s / hs_parts);
    }
    for (ChainRun &r : runs)
    {
        const int64_t keep_global = r.res->global_best_length;
        r.finish();
        if (lockstep)
            r.res->global_best_length = keep_global;
        r.res->seconds = secs;
        r.res->seconds_device = dev_seconds;
        if (r.res->seconds_done == 0.0)
            r.res->seconds_done = secs; // still annealing when the run was stopped
        r.res->seconds_busy = busy_seconds;
        r.res->scored_busy = busy_scored;
    }
    if (n_log)
        *n_log = nlog;
    return LVBGPU_OK;
}
} // namespace

extern "C" int lvbhost_anneal_chains(lvbgpu_ctx *ctx, int32_t R, lvbhost_tree *const *trees, const lvbhost_anneal_params *params,
                                     lvbhost_anneal_result *results, double *log_seconds, int64_t *log_best, int32_t *n_log)
{
    return anneal_chains_run(ctx, R, trees, params, results, log_seconds, log_best, n_log, Clock::now());
}
