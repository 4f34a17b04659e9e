// l3d_hostsort.hpp -- stable host-side ordering of edge lists (the reference sorts std::lists of CLEdge / sparse entries
// with std::list::sort and std::stable_sort: sparsematrix.cc:81-86,157-167, clustering.cc:14; equal keys keep their input
// order).  Bucket by the major key, sort inside the buckets, on several threads.
#pragma once

#include <stdint.h>
#include <sched.h>
#include <unistd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "l3d_options.hpp"

namespace l3d {

// CPUs this process may actually use: affinity mask and cgroup CPU quota (a container can see 256 CPUs and own 16; more busy
// threads than that only get throttled)
inline unsigned usable_cpus()
{
    static const unsigned n = [] {
        unsigned v = std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) v = std::min(v, (unsigned)c); }
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32]; long long period = 0;
            if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
                const long long quota = atoll(q);
                if (quota > 0) v = std::min(v, (unsigned)std::max(1LL, quota / period));
            }
            fclose(f);
        }
        return v;
    }();
    return n;
}

inline unsigned host_threads()         // worker threads of the host-side stages that run alone (finish of compute3Dmodel)
{
    { const int e = tunables().host_threads.load(std::memory_order_relaxed); if (e > 0) return (unsigned)std::min(64, e); }     // L3D_HOST_THREADS
    return std::max(1u, std::min(16u, usable_cpus()));
}

// Worker threads of the host-side stages, kept alive between parallel regions (a finish of compute3Dmodel has a dozen regions of
// a millisecond or two each; creating and joining 15 threads per region cost as much as the regions' work).  One region at a
// time: a region started while another one runs (nested, or from another thread) gets plain threads of its own.
class HostPool {
public:
    // Never destroyed: the (detached) workers may still be parked on its condition variable when the process exits.  A forked
    // child has none of the parent's workers, and the parent's condition variables still count them as waiters (a notify can
    // block on them for ever): the child gets a pool of its own, the inherited one is left alone.
    static HostPool& get()
    {
        static std::atomic<HostPool*> inst{ nullptr };
        HostPool* p = inst.load(std::memory_order_acquire);
        if (!p || p->pid_ != getpid()) {
            static std::atomic_flag busy = ATOMIC_FLAG_INIT;              // (not a mutex: its owner may not exist in a forked child)
            while (busy.test_and_set(std::memory_order_acquire)) std::this_thread::yield();
            p = inst.load(std::memory_order_relaxed);
            if (!p || p->pid_ != getpid()) { p = new HostPool(); p->pid_ = getpid(); inst.store(p, std::memory_order_release); }
            busy.clear(std::memory_order_release);
        }
        return *p;
    }
    void run(unsigned nt, const std::function<void(unsigned)>& fn)
    {
        if (nt <= 1) { fn(0u); return; }
        std::unique_lock<std::mutex> region(region_mu_, std::try_to_lock);
        if (!region.owns_lock()) {
            std::vector<std::thread> th;
            for (unsigned t = 1; t < nt; ++t) th.emplace_back(fn, t);
            fn(0u);
            for (auto& x : th) x.join();
            return;
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            while (n_threads_ + 1 < nt) { const unsigned idx = ++n_threads_; const unsigned long long g = gen_; std::thread([this, idx, g] { loop(idx, g); }).detach(); }
            job_ = &fn; job_threads_ = nt; remaining_ = nt - 1; ++gen_;
        }
        cv_work_.notify_all();
        fn(0u);
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [this] { return remaining_ == 0; });
        job_ = nullptr;
    }
private:
    HostPool() {}
    void loop(unsigned idx, unsigned long long seen)     // seen: the generation current when the thread was created
    {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_work_.wait(lk, [&] { return gen_ != seen; });
            seen = gen_;
            if (idx >= job_threads_) continue;
            const std::function<void(unsigned)>* job = job_;
            lk.unlock();
            (*job)(idx);
            lk.lock();
            if (--remaining_ == 0) cv_done_.notify_one();
        }
    }
    std::mutex region_mu_, mu_;
    std::condition_variable cv_work_, cv_done_;
    unsigned n_threads_ = 0;
    pid_t pid_ = 0;
    const std::function<void(unsigned)>* job_ = nullptr;
    unsigned job_threads_ = 0, remaining_ = 0;
    unsigned long long gen_ = 0;
};

template <class F>
inline void on_threads(unsigned nt, F fn)   // fn(thread) on nt threads, the caller being thread 0
{
    const std::function<void(unsigned)> f = [&fn](unsigned t) { fn(t); };
    HostPool::get().run(nt, f);
}

// Order of n records by (major(i), minor(i), i), major in [0, n_major), minor in [0, n_minor), on nt threads: a counting sort on the major key
// with one private histogram per thread (thread t owns the t-th contiguous slice of the input, so the scatter is stable and
// needs no atomics), then every bucket ordered by (minor, input index) -- skipped when it already is, as for runs of equal
// keys.  bucket_start (optional) receives the n_major + 1 bucket boundaries.
template <class Major, class Minor>
inline void parallel_stable_order(size_t n, size_t n_major, size_t n_minor, Major major, Minor minor, unsigned nt, std::vector<uint32_t>& order,
                                  std::vector<uint32_t>* bucket_start = nullptr)
{
    nt = (unsigned)std::max<size_t>(1, std::min<size_t>(nt, n / 8192 + 1));
    nt = (unsigned)std::max<size_t>(1, std::min<size_t>(nt, ((size_t)1 << 26) / std::max<size_t>(1, n_major)));      // histograms: at most 256 MB
    std::vector<uint32_t> hist((size_t)nt * n_major, 0), start(n_major + 1, 0);
    on_threads(nt, [&](unsigned t) {
        uint32_t* h = hist.data() + (size_t)t * n_major;
        for (size_t i = n * t / nt; i < n * (t + 1) / nt; ++i) ++h[(size_t)major(i)];
    });
    on_threads(nt, [&](unsigned t) {                       // bucket totals
        for (size_t b = n_major * t / nt; b < n_major * (t + 1) / nt; ++b) {
            uint32_t tot = 0;
            for (unsigned q = 0; q < nt; ++q) tot += hist[(size_t)q * n_major + b];
            start[b + 1] = tot;
        }
    });
    for (size_t b = 0; b < n_major; ++b) start[b + 1] += start[b];
    on_threads(nt, [&](unsigned t) {                       // hist[q][b] -> first output slot of thread q in bucket b
        for (size_t b = n_major * t / nt; b < n_major * (t + 1) / nt; ++b) {
            uint32_t run = start[b];
            for (unsigned q = 0; q < nt; ++q) { const uint32_t c = hist[(size_t)q * n_major + b]; hist[(size_t)q * n_major + b] = run; run += c; }
        }
    });
    std::unique_ptr<uint64_t[]> rec(new uint64_t[n]);      // (minor << 32) | input index
    on_threads(nt, [&](unsigned t) {
        uint32_t* h = hist.data() + (size_t)t * n_major;
        for (size_t i = n * t / nt; i < n * (t + 1) / nt; ++i) rec[h[(size_t)major(i)]++] = ((uint64_t)(uint32_t)minor(i) << 32) | (uint32_t)i;
    });
    order.resize(n);
    on_threads(nt, [&](unsigned t) {                       // buckets in slices of about n / nt records
        std::vector<uint64_t> tmp;
        const size_t lo = std::lower_bound(start.begin(), start.end(), (uint32_t)(n * t / nt)) - start.begin();
        const size_t hi = t + 1 == nt ? n_major : std::lower_bound(start.begin(), start.end(), (uint32_t)(n * (t + 1) / nt)) - start.begin();
        for (size_t b = lo; b < hi && b < n_major; ++b) {
            const uint32_t s = start[b], e = start[b + 1];
            uint64_t* r = rec.get() + s;
            const size_t m = e - s;
            if (m > 1 && !std::is_sorted(r, r + m)) {
                if (m < 4096) std::sort(r, r + m);
                else {                                     // a heavy bucket (skewed keys): byte-wise LSD passes over the minor key, stable
                    tmp.resize(m);
                    uint64_t *a = r, *b = tmp.data();
                    for (int shift = 32; shift < 64 && ((uint64_t)(n_minor - 1) >> (shift - 32)) != 0; shift += 8) {
                        size_t cnt[257];
                        memset(cnt, 0, sizeof(cnt));
                        for (size_t i = 0; i < m; ++i) ++cnt[((a[i] >> shift) & 255) + 1];
                        for (int k = 1; k < 256; ++k) cnt[k] += cnt[k - 1];
                        for (size_t i = 0; i < m; ++i) b[cnt[(a[i] >> shift) & 255]++] = a[i];
                        std::swap(a, b);
                    }
                    if (a != r) memcpy(r, a, m * sizeof(uint64_t));
                }
            }
            for (uint32_t k = s; k < e; ++k) order[k] = (uint32_t)rec[k];
        }
    });
    if (bucket_start) bucket_start->swap(start);
}

// monotone map of a float to an unsigned key (-0 == +0; NaNs are not expected in weight lists)
inline uint32_t float_order_key(float w)
{
    if (w == 0.0f) w = 0.0f;
    uint32_t u;
    memcpy(&u, &w, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

}  // namespace l3d
