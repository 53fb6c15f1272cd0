// tests/native/pool_tsan.cpp - the host thread pool and the parallel program build pattern of
// api_batch.cpp (build_into: one private Topology + ProgramBuilder per worker, slices of the batch,
// then a parallel gather) under ThreadSanitizer.  Built and run by tests/test_pool_tsan.py.
// Exit code 0 and "ok" on stdout when the parallel result equals the serial one.
#include <cstdio>
#include <cstring>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "../../lvb_amd/csrc/pool.hpp"
#include "../../lvb_amd/csrc/program.hpp"
#include "../../lvb_amd/csrc/proposals.hpp"

using namespace lvbgpu;

struct Worker
{
    Topology topo;
    ProgramBuilder pb;
    Program prog;
    std::vector<uint32_t> ntok;
};

int main()
{
    const int n = 120, B = 1024, T = 8, rounds = 40;
    Rng rng(7);
    Topology cur;
    random_topology(n, rng, cur);
    Pool pool(T);
    std::vector<Worker> workers(T);
    ProgramBuilder serial(cur.nb);
    for (int round = 0; round < rounds; round++)
    {
        std::vector<Edit> edits;
        std::vector<int32_t> offs{0};
        for (int b = 0; b < B; b++)
        {
            propose(cur, b % 3, rng, edits);
            offs.push_back((int32_t)edits.size());
        }
        pool.run(T, [&](int t) {
            Worker &w = workers[t];
            w.topo = cur; // private copy: build_candidate edits and restores its topology
            w.pb.resize(cur.nb);
            w.prog.toks.clear();
            w.prog.dsts.clear();
            w.ntok.clear();
            std::string why;
            for (int b = B * t / T; b < B * (t + 1) / T; b++)
            {
                const size_t t0 = w.prog.toks.size();
                if (!w.pb.build_candidate(w.topo, edits.data() + offs[b], offs[b + 1] - offs[b], -1, w.prog, &why))
                    std::abort();
                w.ntok.push_back((uint32_t)(w.prog.toks.size() - t0));
            }
        });
        // gather in parallel into one buffer, as the product does
        std::vector<size_t> base(T + 1, 0);
        for (int t = 0; t < T; t++)
            base[t + 1] = base[t] + workers[t].prog.toks.size();
        std::vector<uint32_t> all(base[T]);
        pool.run(T, [&](int t) {
            memcpy(all.data() + base[t], workers[t].prog.toks.data(), workers[t].prog.toks.size() * 4);
        });
        // serial reference
        Program sp;
        std::string why;
        Topology copy = cur;
        for (int b = 0; b < B; b++)
            if (!serial.build_candidate(copy, edits.data() + offs[b], offs[b + 1] - offs[b], -1, sp, &why))
                return 2;
        if (sp.toks != all)
        {
            printf("mismatch in round %d\n", round);
            return 1;
        }
        // move on: accept one candidate so the next round sees another tree
        if (!serial.apply_edits(cur, edits.data() + offs[5], offs[6] - offs[5], -1, &why))
            return 3;
    }
    // the pool's own protocol: runs with fewer tasks than threads (workers that sit a generation out), runs back to
    // back (workers still spinning) and runs after a pause longer than the spin window (workers asleep)
    {
        std::vector<long> hits(T, 0); // task t only ever touches hits[t]
        long expect[8] = {0};
        uint64_t x = 12345;
        for (int run = 0; run < 3000; run++)
        {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            const int active = 1 + (int)((x >> 33) % (uint64_t)T);
            pool.run(active, [&](int t) { hits[t] += t + 1; });
            for (int t = 0; t < active; t++)
                expect[t] += t + 1;
            if (run % 500 == 499)
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
        for (int t = 0; t < T; t++)
            if (hits[t] != expect[t])
            {
                printf("pool protocol: task %d ran %ld, expected %ld\n", t, hits[t], expect[t]);
                return 4;
            }
    }
    printf("ok\n");
    return 0;
}
