// pool.hpp - a few persistent host threads (program building, proposal generation)
#pragma once

#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace lvbgpu
{

// run(active, fn): fn(t) for t in [0, active) on the workers, returns when all are done
class Pool
{
  public:
    explicit Pool(int n)
    {
        for (int t = 0; t < n; t++)
            threads_.emplace_back([this, t] { loop(t); });
    }
    ~Pool()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
            gen_++;
        }
        start_.notify_all();
        for (auto &t : threads_)
            t.join();
    }
    int size() const { return (int)threads_.size(); }
    // run fn(t) for t in [0, active) on the workers and wait for all of them
    void run(int active, const std::function<void(int)> &fn)
    {
        std::unique_lock<std::mutex> g(m_);
        job_ = &fn;
        active_ = active;
        pending_ = active;
        gen_++;
        start_.notify_all();
        done_.wait(g, [this] { return pending_ == 0; });
        job_ = nullptr;
    }

  private:
    void loop(int t)
    {
        uint64_t seen = 0;
        for (;;)
        {
            const std::function<void(int)> *job = nullptr;
            {
                std::unique_lock<std::mutex> g(m_);
                start_.wait(g, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_)
                    return;
                if (t < active_)
                    job = job_;
            }
            if (job)
            {
                (*job)(t);
                std::lock_guard<std::mutex> g(m_);
                if (--pending_ == 0)
                    done_.notify_one();
            }
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable start_, done_;
    const std::function<void(int)> *job_ = nullptr;
    int active_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};


// how many host threads to use (env LVBGPU_THREADS, default 8, never more than the machine has)
inline int host_threads()
{
    const char *e = getenv("LVBGPU_THREADS");
    int n = e ? atoi(e) : 8;
    const int hw = (int)std::thread::hardware_concurrency();
    if (hw > 0 && n > hw)
        n = hw;
    return n < 1 ? 1 : n;
}

} // namespace lvbgpu
