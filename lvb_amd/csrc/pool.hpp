// pool.hpp - a few persistent host threads (program building, proposal generation)
#pragma once
#if defined(__linux__)
#include <sched.h>
#endif

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace lvbgpu
{

// run(active, fn): fn(t) for t in [0, active), returns when all are done.  The caller runs task 0 itself; tasks
// 1 .. active-1 go to the workers.  A search calls this twice per device step, a few hundred microseconds apart,
// and a condition-variable wake-up costs 30-100 us per run: so a worker spins on the job word for a while
// (SPIN_SECONDS) after its last job before it blocks, and the caller spins for the stragglers.
class Pool
{
  public:
    explicit Pool(int n)
    {
        for (int w = 1; w < n; w++)
            threads_.emplace_back([this, w] { loop(w); });
    }
    ~Pool()
    {
        stop_.store(true, std::memory_order_seq_cst);
        state_.fetch_add(1ull << 32, std::memory_order_seq_cst); // a new generation with nobody active
        wake_sleepers();
        for (auto &t : threads_)
            t.join();
    }
    Pool(const Pool &) = delete;
    Pool &operator=(const Pool &) = delete;
    int size() const { return (int)threads_.size() + 1; }
    // run fn(t) for t in [0, active) and wait for all of them (one run at a time: the pool belongs to one context)
    void run(int active, const std::function<void(int)> &fn)
    {
        if (active > size())
            active = size();
        if (active > 1)
        {
            job_ = &fn; // published by the release store of state_ below
            pending_.store(active - 1, std::memory_order_relaxed);
            const uint64_t gen = (state_.load(std::memory_order_relaxed) >> 32) + 1;
            state_.store((gen << 32) | (uint32_t)active, std::memory_order_seq_cst);
            wake_sleepers();
        }
        if (active > 0)
            fn(0);
        if (active > 1)
        {
            for (uint32_t spins = 0; pending_.load(std::memory_order_acquire) != 0; spins++)
                relax(spins);
            job_ = nullptr;
        }
    }

  private:
    static constexpr double SPIN_SECONDS = 300e-6;
    // one step of a spin wait: mostly a pause, now and then the rest of the time slice - on a machine with fewer
    // free cores than threads the one everybody waits for may be the one that is not running
    static void relax(uint32_t spins)
    {
        if ((spins & 63u) == 63u)
            std::this_thread::yield();
        else
        {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
    }
    void wake_sleepers()
    {
        // a worker counts itself in under the mutex before it re-checks the generation, so either it sees the
        // new one or we see it here
        if (sleepers_.load(std::memory_order_seq_cst) > 0)
        {
            std::lock_guard<std::mutex> g(m_);
            start_.notify_all();
        }
    }
    void loop(int w)
    {
        uint64_t seen = 0; // generation of the last job word this worker acted on
        uint32_t idle_spins = 0;
        auto idle_since = std::chrono::steady_clock::now();
        for (;;)
        {
            uint64_t st = state_.load(std::memory_order_acquire);
            if ((st >> 32) == seen)
            {
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - idle_since).count() < SPIN_SECONDS)
                {
                    relax(idle_spins++);
                    continue;
                }
                std::unique_lock<std::mutex> g(m_);
                sleepers_.fetch_add(1, std::memory_order_seq_cst);
                start_.wait(g, [&] { return (state_.load(std::memory_order_seq_cst) >> 32) != seen; });
                sleepers_.fetch_sub(1, std::memory_order_seq_cst);
                continue;
            }
            seen = st >> 32;
            if (stop_.load(std::memory_order_seq_cst))
                return;
            // (generation, active) come from ONE word: a worker that is not part of this generation never touches
            // job_ or pending_, and one that is keeps run() from returning until it has finished
            if (w < (int)(uint32_t)st)
            {
                (*job_)(w);
                pending_.fetch_sub(1, std::memory_order_release);
            }
            idle_since = std::chrono::steady_clock::now();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable start_;
    const std::function<void(int)> *job_ = nullptr;
    std::atomic<uint64_t> state_{0}; // generation << 32 | tasks of that generation
    std::atomic<int> pending_{0};
    std::atomic<int> sleepers_{0};
    std::atomic<bool> stop_{false};
};


// how many host threads to use: env LVBGPU_THREADS, else this process's share of the cores it may run on - its affinity
// mask divided by LOCAL_WORLD_SIZE (one process per GPU: eight ranks with sixteen workers each and no affinity would
// fight over the host) - at least 2, at most 16 (the share of host cores one GPU of an 8-GPU node comes with); never more
// than the process may run on
inline int host_threads()
{
    int allowed = (int)std::thread::hardware_concurrency();
#if defined(__linux__)
    {
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0)
            allowed = CPU_COUNT(&set);
    }
#endif
    const char *e = getenv("LVBGPU_THREADS");
    int n;
    if (e)
        n = atoi(e);
    else
    {
        const char *lw = getenv("LOCAL_WORLD_SIZE");
        const int ranks_here = lw && atoi(lw) > 0 ? atoi(lw) : 1;
        n = allowed > 0 ? allowed / ranks_here : 16;
        n = n < 2 ? 2 : (n > 16 ? 16 : n);
    }
    if (allowed > 0 && n > allowed)
        n = allowed;
    return n < 1 ? 1 : n;
}

} // namespace lvbgpu
