#include "../lvb_amd/csrc/pool.hpp"
#include <chrono>
#include <cstdio>
#include <atomic>
#include <algorithm>
#include <vector>
int main(int argc, char **argv)
{
    int n = argc > 1 ? atoi(argv[1]) : 8;
    lvbgpu::Pool pool(n);
    std::atomic<long> sink{0};
    for (int work : {0, 20000})
    {
        std::vector<double> ts;
        for (int r = 0; r < 3000; r++)
        {
            auto a = std::chrono::steady_clock::now();
            pool.run(n, [&](int t) { long s = 0; for (int i = 0; i < work; i++) s += i * t; sink += s; });
            ts.push_back(std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count() * 1e6);
        }
        std::sort(ts.begin(), ts.end());
        double sum = 0; for (double v : ts) sum += v;
        printf("n=%d work %d: mean %.1f median %.1f p90 %.1f p99 %.1f max %.1f us\n", n, work, sum / ts.size(), ts[ts.size()/2], ts[ts.size()*9/10], ts[ts.size()*99/100], ts.back());
    }
}
