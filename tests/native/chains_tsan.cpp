// tests/native/chains_tsan.cpp - lvbhost_anneal_chains with runs of acceptances under ThreadSanitizer (CPU test tier).
// The hot chains' candidates are drawn, consumed and followed on several host threads (lvbgpu_parallel_for); the scorer
// is the TEST DOUBLE (tests/cpu_double/lvbgpu_double.c: the oracle behind the lvbgpu_* calls the host makes - test tier
// only, never part of the product).  What must hold: no data race between the chains' tasks, and every chain ends where
// it ends with one move per step.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/lvbhost.h"

extern "C"
{
#include "../../oracle/fitch_oracle.h"
    lvbgpu_ctx *lvbgpu_double_new(long n, long nwords, const uint64_t *enc);
    void lvbgpu_double_free(lvbgpu_ctx *c);
}

int main()
{
    const int n = 10, m = 96, R = 5;
    // a small tree-like alignment: taxon i copies taxon (i - 1) / 2 with a tenth of its sites changed
    std::mt19937 rng(7);
    std::vector<std::string> rows((size_t)n, std::string((size_t)m, 'A'));
    const char acgt[] = "ACGT";
    for (int j = 0; j < m; j++)
        rows[0][(size_t)j] = acgt[rng() % 4];
    for (int i = 1; i < n; i++)
        for (int j = 0; j < m; j++)
            rows[(size_t)i][(size_t)j] = rng() % 10 == 0 ? acgt[rng() % 4] : rows[(size_t)(i - 1) / 2][(size_t)j];
    std::vector<const char *> ptr;
    for (auto &r : rows)
        ptr.push_back(r.c_str());
    const long nwords = lvbo_words_per_row(m);
    std::vector<uint64_t> enc((size_t)n * (size_t)nwords);
    for (int i = 0; i < n; i++)
        if (lvbo_encode_row(ptr[(size_t)i], m, nwords, enc.data() + (size_t)i * (size_t)nwords) != 0)
            return 2;
    const int64_t min_len = lvbhost_min_tree_length(n, m, ptr.data());

    auto run = [&](std::vector<lvbhost_anneal_result> &res, int run_levels) {
        lvbgpu_ctx *ctx = lvbgpu_double_new(n, nwords, enc.data());
        std::vector<lvbhost_tree *> trees;
        std::vector<lvbhost_anneal_params> pars((size_t)R);
        for (int c = 0; c < R; c++)
        {
            trees.push_back(lvbhost_tree_random(n, 100 + (uint64_t)c));
            lvbhost_anneal_defaults(&pars[(size_t)c]);
            pars[(size_t)c].seed = 900 + (uint64_t)c;
            pars[(size_t)c].algorithm = 1;
            pars[(size_t)c].batch = 32;
            pars[(size_t)c].t0 = c == 0 ? 0.0 : 0.0005; // one chain estimates its starting temperature first
            pars[(size_t)c].min_len_tree = min_len;
            pars[(size_t)c].max_proposals = 120;
            pars[(size_t)c].log_cap = 32;
            pars[(size_t)c].run_levels = run_levels;
        }
        std::vector<double> secs(32);
        std::vector<int64_t> best(32);
        int32_t nlog = 0;
        res.assign((size_t)R, lvbhost_anneal_result{});
        const int rc = lvbhost_anneal_chains(ctx, R, trees.data(), pars.data(), res.data(), secs.data(), best.data(), &nlog);
        for (auto *t : trees)
            lvbhost_tree_free(t);
        lvbgpu_double_free(ctx);
        return rc;
    };
    // runs of acceptances: the hot chains' candidates are drawn and consumed on several threads (lvbgpu_parallel_for):
    // no race between the chains' tasks, and a chain ends where it ends with one move per step
    std::vector<lvbhost_anneal_result> runs1, runs3;
    if (run(runs1, 1) != 0 || run(runs3, 3) != 0)
        return 5;
    int64_t host_steps = 0;
    for (int c = 0; c < R; c++)
    {
        host_steps += runs3[(size_t)c].host_steps;
        if (runs1[(size_t)c].best_length != runs3[(size_t)c].best_length || runs1[(size_t)c].final_length != runs3[(size_t)c].final_length ||
            runs1[(size_t)c].consumed != runs3[(size_t)c].consumed || runs1[(size_t)c].accepted != runs3[(size_t)c].accepted ||
            runs1[(size_t)c].topologies != runs3[(size_t)c].topologies)
        {
            printf("chain %d differs between run lengths\n", c);
            return 6;
        }
    }
    if (host_steps == 0)
    {
        printf("no host-drawn step happened\n");
        return 7;
    }
    printf("ok\n");
    return 0;
}
