// CPU test of plastid_amd/csrc/host_util.h (compiled and run by tests/test_host_logic.py; test infrastructure).
#include <atomic>
#include <cstdio>
#include <random>
#include "host_util.h"

int main() {
    long bad = 0;
    // gallop_lower_bound == std::lower_bound for every hint
    std::mt19937_64 rng(5);
    for (int it = 0; it < 100000; ++it) {
        const size_t n = rng() % 40;
        std::vector<int> v(n);
        for (auto &x : v) x = (int)(rng() % 30);
        std::sort(v.begin(), v.end());
        const int key = (int)(rng() % 34) - 2;
        const size_t hint = n ? rng() % (n + 3) : 0;
        const size_t got = gallop_lower_bound(n, hint, [&](size_t i) { return v[i] < key; });
        const size_t want = (size_t)(std::lower_bound(v.begin(), v.end(), key) - v.begin());
        if (got != want) ++bad;
    }
    printf("gallop_lower_bound: bad %ld\n", bad);
    // parallel_chunks covers [0, n) exactly once, whatever the width (regions served by the pool)
    for (int r = 0; r < 4000; ++r) {
        const int T = 1 + r % 17;
        const int64_t n = r % 3 == 0 ? 5 : 1000 + r;
        std::vector<int> hit((size_t)n, 0);
        parallel_chunks(n, T, [&](int, int64_t b, int64_t e) { for (int64_t i = b; i < e; ++i) hit[(size_t)i] += 1; });
        for (int v : hit) if (v != 1) ++bad;
    }
    // nested regions and concurrent callers (the pool is taken: those regions spawn their own threads)
    std::atomic<long> total{0};
    std::vector<std::thread> callers;
    for (int c = 0; c < 4; ++c) callers.emplace_back([&] {
        for (int r = 0; r < 500; ++r)
            parallel_chunks(64, 8, [&](int, int64_t b, int64_t e) {
                parallel_chunks(e - b, 3, [&](int, int64_t b2, int64_t e2) { total += e2 - b2; });
            });
    });
    for (auto &t : callers) t.join();
    if (total.load() != 4L * 500 * 64) ++bad;
    printf("parallel_chunks: total %ld (want %ld)\n", total.load(), 4L * 500 * 64);
    // PodVec: resize keeps what was written, push_back works
    PodVec<int> pv;
    pv.resize(1000);
    for (int i = 0; i < 1000; ++i) pv[(size_t)i] = i;
    pv.push_back(7);
    pv.resize(2000);
    for (int i = 0; i < 1000; ++i) if (pv[(size_t)i] != i) ++bad;
    if (pv[1000] != 7) ++bad;
    // scan_contigs == a record-by-record walk: bounds of the sorted prefix, first record out of order / out of range
    for (int it = 0; it < 3000; ++it) {
        const int ntid = 1 + (int)(rng() % 9);
        const int64_t n = (int64_t)(rng() % (it % 50 == 0 ? 30000 : 300));
        std::vector<int32_t> tid((size_t)n);
        int32_t cur = 0;
        for (auto &x : tid) { if (rng() % 17 == 0) cur = std::min<int32_t>(ntid - 1, cur + (int32_t)(rng() % 3)); x = cur; }
        for (int d = (int)(rng() % 3); d > 0 && n > 0; --d)   // defects: a contig out of range, a step back
            tid[(size_t)(rng() % (uint64_t)n)] = (rng() & 1) ? (int32_t)(rng() % 5) - 2 : ntid + (int32_t)(rng() % 2);
        int64_t want_ok = n;
        for (int64_t i = 0; i < n; ++i)
            if (tid[(size_t)i] < 0 || tid[(size_t)i] >= ntid || (i > 0 && tid[(size_t)i] < tid[(size_t)i - 1])) { want_ok = i; break; }
        std::vector<int64_t> want((size_t)ntid + 1, want_ok), got;
        for (int t = 0; t <= ntid; ++t)
            for (int64_t i = 0; i < want_ok; ++i)
                if (tid[(size_t)i] >= t) { want[(size_t)t] = i; break; }
        const int64_t got_ok = scan_contigs(tid.data(), n, ntid, 1 + it % 7, got);
        if (got_ok != want_ok || got != want) ++bad;
    }
    printf("scan_contigs: bad %ld\n", bad);
    // choose_window: the window sizes the published numbers were measured with, and the knob's limits
    {
        int64_t g = 0;
        if (choose_window(1, 2, 21031, 33373947ull, 0, &g) != 2048 || g != 3072) ++bad;          // C2 / C3: dense, stranded
        if (choose_window(1, 2, 479339, 91633228ull, 0, &g) != 768) ++bad;                       // C4: exons, a quarter of a window or less
        if (choose_window(1, 1, 10, 100000ull, 0, &g) != 4096 || g != 6144) ++bad;               // one strand mode: capped at 4 096
        if (choose_window(11, 2, 479339, 91633228ull, 0, &g) != 768 || g != 837) ++bad;          // C5: 16-bit bins, 36 KiB, multiples of 256
        if (choose_window(36, 2, 100, 100000ull, 0, &g) != 256) ++bad;                           // many rows: never below 256
        if (choose_window(3, 1, 100, 100000ull, 0, &g) != 4096) ++bad;
        if (choose_window(11, 2, 100, 100000ull, 512, &g) != 512) ++bad;                         // PC_TILE_G within twice the budget
        if (choose_window(11, 2, 100, 100000ull, 4096, &g) != 768) ++bad;                        // ... and beyond it: ignored
        if (choose_window(1, 2, 0, 0ull, 0, &g) != 2048) ++bad;                                  // no intervals: not sparse
    }
    printf("choose_window: bad %ld\n", bad);
    printf("host_util: %s\n", bad ? "FAILED" : "ok");
    return bad != 0;
}
