// host_pipe.h -- the host-buffer entry points' way into the device (sdhip_*_push / _flush / _pull: what the pipeline modules call): samples are
// gathered into one of TWO pinned staging buffers by several copy threads, and a full buffer is handed to a worker thread that ships and
// processes it (H2D, the engine's kernels, D2H of the results) while the caller fills the other one. Round 3 had one pinned buffer filled by a
// single memcpy and processed synchronously: 13 GB/s; a single core copies 10-13 GB/s, PCIe 5 x16 carries ~55.
#pragma once
#include "common.h"

#include <algorithm>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace sdhip
{
    // memcpy split over a small persistent pool (created on first use; SDHIP_COPY_THREADS, default 8 or half the cores): the staging copy is
    // memory-bound, a handful of cores saturate what one NUMA node gives
    class CopyPool
    {
        std::vector<std::thread> th;
        std::mutex mu, call_mu; // call_mu: one copy() at a time (the caller's staging copy and the worker's result copy may meet)
        std::condition_variable cv_go, cv_done;
        struct Job
        {
            char *dst = nullptr;
            const char *src = nullptr;
            size_t bytes = 0;
        };
        std::vector<Job> jobs;
        unsigned long long gen = 0;
        int pending = 0;
        bool quit = false;
        void run(int i)
        {
            unsigned long long seen = 0;
            for (;;)
            {
                Job j;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv_go.wait(lk, [&] { return quit || gen != seen; });
                    if (quit)
                        return;
                    seen = gen;
                    j = jobs[i];
                }
                if (j.bytes)
                    memcpy(j.dst, j.src, j.bytes);
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (--pending == 0)
                        cv_done.notify_all();
                }
            }
        }

      public:
        explicit CopyPool(int n)
        {
            jobs.resize(n);
            for (int i = 0; i < n; i++)
                th.emplace_back([this, i] { run(i); });
        }
        ~CopyPool()
        {
            {
                std::lock_guard<std::mutex> lk(mu);
                quit = true;
            }
            cv_go.notify_all();
            for (auto &t : th)
                t.join();
        }
        void copy(void *dst, const void *src, size_t bytes)
        {
            const int n = (int)th.size();
            if (bytes < ((size_t)2 << 20) || n <= 1)
            {
                memcpy(dst, src, bytes);
                return;
            }
            std::lock_guard<std::mutex> one(call_mu);
            std::unique_lock<std::mutex> lk(mu);
            const size_t per = ((bytes + n - 1) / n + 4095) & ~(size_t)4095;
            for (int i = 0; i < n; i++)
            {
                const size_t o = std::min(bytes, per * i), e = std::min(bytes, per * (i + 1));
                jobs[i] = Job{(char *)dst + o, (const char *)src + o, e - o};
            }
            pending = n;
            gen++;
            cv_go.notify_all();
            cv_done.wait(lk, [&] { return pending == 0; });
        }
        static CopyPool &get()
        {
            static CopyPool *p = [] {
                int n = 0;
                if (const char *v = getenv("SDHIP_COPY_THREADS"))
                    n = atoi(v);
                if (n <= 0)
                    n = (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency() / 2));
                return new CopyPool(n); // lives as long as the process (its threads sleep on a condition variable)
            }();
            return *p;
        }
    };

    class HostPipe
    {
      public:
        // thread A: ship a staged batch into device slot `slot` (H2D, synchronous); thread B: process slot `slot` and queue the results. While B works on
        // batch k, A ships batch k + 1 and the caller stages batch k + 2.
        using Ship = std::function<void(const uint8_t *pinned, size_t bytes, int fmt, int slot)>;
        using Compute = std::function<void(size_t bytes, int fmt, int slot)>;

      private:
        PinBuf<uint8_t> buf[2];
        size_t fill = 0; // bytes gathered in buf[cur]
        int cur = 0, fmt = 0;
        bool busy[2] = {false, false};      // staging buffer handed to thread A
        bool slot_busy[2] = {false, false}; // device slot shipped, not yet processed
        std::vector<int> order;             // submitted staging buffers, oldest first
        struct Job
        {
            size_t bytes;
            int fmt, slot;
        };
        std::vector<Job> jobs; // shipped, waiting for thread B
        bool computing = false;
        size_t sub_bytes[2] = {0, 0};
        int sub_fmt[2] = {0, 0};
        int next_slot = 0;
        std::mutex mu;
        std::condition_variable cv;
        std::thread ta, tb;
        bool quit = false;
        std::exception_ptr err;
        Ship ship;
        Compute compute;
        void fail()
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!err)
                err = std::current_exception();
        }
        void loop_a()
        {
            for (;;)
            {
                int b, s;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return quit || !order.empty(); });
                    if (order.empty())
                        return;
                    b = order.front();
                    s = next_slot;
                    cv.wait(lk, [&] { return quit || !slot_busy[s]; });
                    if (slot_busy[s])
                        return;
                    slot_busy[s] = true;
                    next_slot ^= 1;
                }
                bool ok = true;
                try
                {
                    ship(buf[b].p, sub_bytes[b], sub_fmt[b], s);
                }
                catch (...)
                {
                    ok = false;
                    fail();
                }
                {
                    std::lock_guard<std::mutex> lk(mu);
                    order.erase(order.begin());
                    busy[b] = false;
                    if (ok)
                        jobs.push_back(Job{sub_bytes[b], sub_fmt[b], s});
                    else
                        slot_busy[s] = false;
                }
                cv.notify_all();
            }
        }
        void loop_b()
        {
            for (;;)
            {
                Job j;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return quit || !jobs.empty(); });
                    if (jobs.empty())
                        return;
                    j = jobs.front();
                    jobs.erase(jobs.begin());
                    computing = true;
                }
                try
                {
                    bool bad;
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        bad = (bool)err;
                    }
                    if (!bad)
                        compute(j.bytes, j.fmt, j.slot);
                }
                catch (...)
                {
                    fail();
                }
                {
                    std::lock_guard<std::mutex> lk(mu);
                    slot_busy[j.slot] = false;
                    computing = false;
                }
                cv.notify_all();
            }
        }
        void submit()
        { // buf[cur] goes to the shipping thread, the caller moves on to the other one (waiting for it if it is still being shipped)
            {
                std::lock_guard<std::mutex> lk(mu);
                busy[cur] = true;
                sub_bytes[cur] = fill;
                sub_fmt[cur] = fmt;
                order.push_back(cur);
            }
            cv.notify_all();
            cur ^= 1;
            fill = 0;
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !busy[cur]; });
        }
        void rethrow()
        {
            std::exception_ptr e;
            {
                std::lock_guard<std::mutex> lk(mu);
                e = err;
                err = nullptr;
            }
            if (e)
                std::rethrow_exception(e);
        }

      public:
        HostPipe(Ship s, Compute c) : ship(std::move(s)), compute(std::move(c)) {}
        ~HostPipe()
        {
            {
                std::lock_guard<std::mutex> lk(mu);
                quit = true;
            }
            cv.notify_all();
            if (ta.joinable())
                ta.join();
            if (tb.joinable())
                tb.join();
        }
        size_t pending_bytes() const { return fill; }
        // append `bytes` of samples in format f; a buffer is shipped when it holds batch_bytes
        void push(const void *src, size_t bytes, int f, size_t batch_bytes)
        {
            rethrow();
            if (fill != 0 && f != fmt)
                throw HipError("baseband format changed mid-stream");
            fmt = f;
            if (!ta.joinable())
            {
                ta = std::thread([this] { loop_a(); });
                tb = std::thread([this] { loop_b(); });
            }
            const uint8_t *s = (const uint8_t *)src;
            while (bytes)
            {
                const size_t take = std::min(bytes, batch_bytes - fill);
                if (fill + take > buf[cur].cap)
                { // grow, keeping what is gathered (the pinned buffers reach batch_bytes only for callers that push that much)
                    PinBuf<uint8_t> bigger;
                    bigger.reserve(std::min(batch_bytes, std::max<size_t>(2 * (fill + take), (size_t)1 << 22)));
                    if (fill)
                        memcpy(bigger.p, buf[cur].p, fill);
                    std::swap(bigger.p, buf[cur].p);
                    std::swap(bigger.cap, buf[cur].cap);
                }
                CopyPool::get().copy(buf[cur].p + fill, s, take);
                fill += take;
                s += take;
                bytes -= take;
                if (fill >= batch_bytes)
                    submit();
            }
        }
        // ship what is gathered and wait until both threads have nothing left
        void flush()
        {
            if (fill)
                submit();
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return order.empty() && jobs.empty() && !computing; });
            }
            rethrow();
        }
    };
} // namespace sdhip
