// host_util.h -- host-side helpers of the counting engine that do not touch HIP: the worker pool behind
// parallel_chunks, the galloping lower bound of the plan build, a vector without zero-fill, the host's look at the
// contig column of caller-owned records (scan_contigs), the window size of a plan (choose_window).  Plain C++17, so that
// tests/test_host_logic.py can compile tests/host_util_test.cpp against it on a machine without a GPU.
#pragma once
#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>
#include <unistd.h>

// std::vector whose resize() leaves trivially constructible elements uninitialised: the plan's tables are sized
// once and then filled by all threads, and a serial zero-fill of tens of megabytes (plus the page faults it takes on
// one thread) cost more than the fill itself.
template <typename T> struct NoInitAlloc : std::allocator<T> {
    template <typename U> struct rebind { typedef NoInitAlloc<U> other; };
    NoInitAlloc() = default;
    template <typename U> NoInitAlloc(const NoInitAlloc<U> &) {}
    template <typename U> void construct(U *q) { ::new ((void *)q) U; }
    template <typename U, typename... A> void construct(U *q, A &&...a) { ::new ((void *)q) U(std::forward<A>(a)...); }
};
template <typename T> using PodVec = std::vector<T, NoInitAlloc<T>>;

// First index in [0, n) for which `before(idx)` is false (before() is monotone: true ... true false ... false),
// searched outward from `hint`: consecutive queries of a plan (the exons of a chain) land next to each other, and a
// bisection over a few hundred thousand entries costs ~18 cache misses where the gallop costs two or three.
template <typename F> static size_t gallop_lower_bound(size_t n, size_t hint, F before) {
    if (n == 0) return 0;
    if (hint >= n) hint = n - 1;
    size_t lo, hi;   // the answer lies in [lo, hi]: before(lo - 1) holds (or lo == 0), before(hi) does not (or hi == n)
    if (before(hint)) {
        size_t step = 1;
        lo = hint + 1;
        hi = lo;
        while (hi < n && before(hi)) { lo = hi + 1; hi += step; step <<= 1; }
        if (hi > n) hi = n;
    } else {
        size_t step = 1;
        lo = hi = hint;
        while (lo > 0) {
            const size_t probe = lo > step ? lo - step : 0;
            if (before(probe)) { lo = probe + 1; break; }
            lo = hi = probe;
            step <<= 1;
        }
    }
    while (lo < hi) {
        const size_t mid = lo + (hi - lo) / 2;
        if (before(mid)) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// A process-wide pool of worker threads for the host passes (staging, plan build): a 16-thread region costs ~0.5 ms to
// spawn and join, and one staging call runs some seventy of them.  One region at a time uses the pool (the caller is
// worker 0); a region that finds it taken -- another engine staging on another host thread, a nested region -- spawns
// its own threads as before.
class WorkerPool {
    std::mutex gate_;                       // held for the duration of a region
    std::mutex m_;
    std::condition_variable start_, done_;
    std::vector<std::thread> workers_;      // worker k has index k + 1
    const std::function<void(int)> *job_ = nullptr;
    int njobs_ = 0;                         // indices [1, njobs_) take part in the current region
    int pending_ = 0;
    uint64_t generation_ = 0;
    bool stop_ = false;
    const pid_t pid_ = getpid();
    void loop(int idx) {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            start_.wait(lk, [&] { return stop_ || generation_ != seen; });
            if (stop_) return;
            seen = generation_;
            if (idx >= njobs_) continue;
            const std::function<void(int)> *job = job_;
            lk.unlock();
            (*job)(idx);
            lk.lock();
            if (--pending_ == 0) done_.notify_one();
        }
    }
public:
    // runs f(0) ... f(n - 1), f(0) on the calling thread; false (nothing done) when the pool is in use
    bool run(int n, const std::function<void(int)> &f) {
        if (getpid() != pid_) return false;   // a forked child has the pool's bookkeeping but not its threads
        // a region nested inside a job of this pool: worker 0 of the outer region is the thread that owns gate_, and
        // try_lock on a mutex the caller already owns is undefined -- every thread that runs a job says so itself
        static thread_local bool in_region = false;
        if (in_region) return false;
        std::unique_lock<std::mutex> region(gate_, std::try_to_lock);
        if (!region.owns_lock()) return false;
        struct Mark { bool &f; explicit Mark(bool &x) : f(x) { f = true; } ~Mark() { f = false; } } mark(in_region);
        {
            std::lock_guard<std::mutex> lk(m_);
            while ((int)workers_.size() < n - 1) {
                const int idx = (int)workers_.size() + 1;
                workers_.emplace_back([this, idx] { loop(idx); });
            }
            job_ = &f;
            njobs_ = n;
            pending_ = n - 1;
            ++generation_;
        }
        start_.notify_all();
        f(0);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return pending_ == 0; });
        job_ = nullptr;
        njobs_ = 0;
        return true;
    }
};
// (never destroyed: its threads sleep until the process ends -- no joins during static destruction)
static WorkerPool &worker_pool() { static WorkerPool *p = new WorkerPool; return *p; }

// fn(thread index, begin, end) over [0, n) cut into `nthreads` contiguous chunks
template <typename F> static void parallel_chunks(int64_t n, int nthreads, F fn) {
    if (nthreads <= 1 || n <= 0) { fn(0, (int64_t)0, n); return; }
    const int64_t chunk = (n + nthreads - 1) / nthreads;
    auto part = [&](int t) {
        const int64_t b = std::min<int64_t>(n, (int64_t)t * chunk), e_ = std::min<int64_t>(n, b + chunk);
        fn(t, b, e_);
    };
    if (worker_pool().run(nthreads, part)) return;
    std::vector<std::thread> th;
    th.reserve((size_t)nthreads);
    for (int t = 0; t < nthreads; ++t) th.emplace_back(part, t);
    for (auto &x : th) x.join();
}

// The contig column of caller-owned records, examined where it is: `bounds[t]` = first record of contig t (t = 0 ..
// ntid) and, returned, the first record whose contig is out of range or lower than its predecessor's (n: none; the
// bounds then describe the records before it).  Sorted, the column changes value at most ntid times: the pass is a
// streaming comparison of neighbours, and only a block that holds a change is looked at record by record.
static int64_t scan_contigs(const int32_t *tid, int64_t n, int32_t ntid, int threads, std::vector<int64_t> &bounds) {
    bounds.assign((size_t)ntid + 1, 0);
    if (n <= 0) return 0;
    struct Part { std::vector<std::pair<int64_t, int32_t>> changes; int64_t bad = INT64_MAX; };
    const int T = std::max(1, threads);
    std::vector<Part> parts((size_t)T);
    parallel_chunks(n, T, [&](int t, int64_t b, int64_t en) {
        Part &pt = parts[(size_t)t];
        int64_t i = b;
        if (i == 0 && i < en) {
            if ((uint32_t)tid[0] >= (uint32_t)ntid) { pt.bad = 0; return; }
            pt.changes.emplace_back(0, tid[0]);
            i = 1;
        }
        constexpr int64_t kBlock = 4096;
        while (i < en) {
            const int64_t e2 = std::min(en, i + kBlock);
            uint32_t diff = 0;
            for (int64_t j = i; j < e2; ++j) diff |= (uint32_t)(tid[j] ^ tid[j - 1]);
            if (diff)
                for (int64_t j = i; j < e2; ++j)
                    if (tid[j] != tid[j - 1]) {
                        if (tid[j] < tid[j - 1] || tid[j] < 0 || tid[j] >= ntid) { pt.bad = j; return; }
                        pt.changes.emplace_back(j, tid[j]);
                    }
            i = e2;
        }
    });
    int64_t bad = INT64_MAX;
    for (const auto &pt : parts) bad = std::min(bad, pt.bad);
    const int64_t n_ok = std::min(n, bad);
    int32_t last = -1;   // bounds are written up to this contig
    for (const auto &pt : parts)
        for (const auto &c : pt.changes) {
            if (c.first >= n_ok) break;
            for (int32_t t = last + 1; t <= c.second; ++t) bounds[(size_t)t] = c.first;
            last = c.second;
        }
    for (int32_t t = last + 1; t <= ntid; ++t) bounds[(size_t)t] = n_ok;
    return n_ok;
}

// Window size G of a plan: the positions a workgroup of k_hist_point bins in LDS (per strand mode and row).
//   one row     32-bit bins, 24 KiB of them (six workgroups per CU), a power of two; a SPARSE annotation (queried intervals
//               a quarter of a window long on average: exons of a human-scale genome) takes 3/8 of that -- fewer bins
//               to clear and read per exon, windows that start closer to their records (C4: 2 048 -> 1.01, 1 024 -> 0.94,
//               768 -> 0.90 - 0.92 ms; below 768 twice the time; the dense C2 is flat from 1 536 to 3 072 and keeps 2 048)
//   more rows   (the stratified rule) 16-bit bins, 36 KiB of them, any multiple of 256: a plan of 11 rows gets 768 positions
//               (C5: 512 -> 3.38, 640 -> 3.27, 768 -> 3.21, 896 -> 3.21 ms on one box: a third fewer windows outweigh the
//               fifth workgroup per CU they cost)
// `knob`: PC_TILE_G, any multiple of 256 up to twice the budget.
static int choose_window(int rows, int nmodes, unsigned long long n_iv, unsigned long long iv_len, int knob, int64_t *budget) {
    const int64_t bin_bytes = rows > 1 ? 2 : 4;
    const int64_t g = ((rows > 1 ? 36 : 24) * 1024) / (bin_bytes * nmodes * rows);
    int G = 256;
    if (rows > 1) G = (int)std::max<int64_t>(256, std::min<int64_t>(4096, g / 256 * 256));
    else {
        while (G * 2 <= g && G * 2 <= 4096) G *= 2;
        if (G >= 2048 && n_iv > 0 && iv_len * 4 < n_iv * (unsigned long long)G) G = G / 8 * 3;
    }
    if (knob && knob <= 2 * g) G = knob;
    *budget = g;
    return G;
}
