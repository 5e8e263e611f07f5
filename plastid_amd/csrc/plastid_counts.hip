// plastid_counts.hip -- host side of the C ABI declared in include/plastid_counts.h.
//
// Owns device memory, builds the interval plan (islands -> genome tiles ->
// pieces), launches the kernels of pc_kernels.hip.h on the engine's stream and
// times them with HIP events.  There is no CPU counting path in this file: every
// count comes out of a HIP kernel.
//
// Host-side structure, in file order:
//   DevPool / DevBuf / PinnedBuf   device blocks recycled per engine, one page-locked buffer: a plan of
//                                  one short segment costs API calls, not bytes
//   TransferRing / scan_contigs      the caller's columns cross PCIe through a ring of page-locked pieces; the
//                                  contig column stays on the host and becomes ntid + 1 record bounds
//   pc_add_alignment_file          staging: columns to HBM, then kernels only (stage_kernels.hip.h: validation,
//                                  8-byte records, run stream, statistics; pc_kernels.hip.h: record stream,
//                                  side lists, linear-index tables)
//   pc_plan_create                 segments -> islands -> windows -> output pieces, all tables of a
//                                  plan in one device block
//   pc_count                       k_tile_ranges -> k_hist_point (two classes, two streams) ->
//                                  k_gather_split, or the three center kernels + k_gather
//   pc_rle / pc_total / pc_mapped_reads / pc_warn_flags   consumers of a finished count
#include "pc_kernels.hip.h"
#include "plan_kernels.hip.h"
#include "stage_kernels.hip.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <chrono>
#include <future>
#include <functional>
#include <condition_variable>
#include <map>
#include <mutex>
#include <unistd.h>
#include <mutex>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <cerrno>
#include <thread>
#include <vector>

#include "plastid_counts.h"
#include "host_util.h"

using namespace pc;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t err__ = (expr);                                                                 \
        if (err__ != hipSuccess)                                                                   \
            return fail(err__ == hipErrorOutOfMemory ? PC_ERR_NOMEM : PC_ERR_HIP, "%s failed: %s (%s:%d)", \
                        #expr, hipGetErrorString(err__), __FILE__, __LINE__);                      \
    } while (0)

// Device blocks that outlive their engine: an engine that goes away hands the idle blocks of its pool to this
// process-wide reservoir (per device, by size) instead of freeing them, and the pool of a later engine looks here
// before it allocates.  Engines come and go -- one per BAMGenomeArray, one per config in bench.py -- and on some
// boxes of the pool a hipMalloc that follows the hipFree of tens of GB takes SECONDS (measured round 5: 1.5 s and 2.1 s
// for the first large buffer of a staging call right after an engine of the same size was destroyed; milliseconds when
// nothing had been freed).  Only blocks of a destroyed engine get here -- its streams are drained by then, so nothing
// is in flight on them.
// What is kept is bounded: PC_POOL_RESERVOIR_GB per device (default 4: the tables of a few plans and a small file; a
// process that cycles through engines of tens of GB -- bench.py -- opts in to more), and when the LAST engine of a device
// is destroyed everything above that default is freed, so that another library in the process (torch, cupy) finds the
// HBM this one no longer uses.  pc_release_cached_memory frees all of it.
struct BigReservoir {
    static constexpr size_t kDefaultLimit = (size_t)4 << 30;
    std::mutex m;
    struct PerDevice { std::map<size_t, std::vector<void *>> big; size_t cached = 0; int engines = 0; };
    std::map<int, PerDevice> dev;   // keyed by the device id itself (no folding of ids onto a fixed table)
    size_t limit = kDefaultLimit;
    BigReservoir() { if (const char *env = getenv("PC_POOL_RESERVOIR_GB")) limit = (size_t)std::max(0ll, atoll(env)) << 30; }
    static BigReservoir &get() { static BigReservoir *r = new BigReservoir; return *r; }   // (never destroyed: no hipFree during static destruction)
    void *take(int device, size_t rounded) {
        std::lock_guard<std::mutex> g(m);
        PerDevice &d = dev[device];
        auto it = d.big.find(rounded);
        if (it == d.big.end() || it->second.empty()) return nullptr;
        void *p = it->second.back();
        it->second.pop_back();
        d.cached -= rounded;
        return p;
    }
    bool give(int device, void *p, size_t rounded) {
        std::lock_guard<std::mutex> g(m);
        PerDevice &d = dev[device];
        if (d.cached + rounded > limit) return false;
        d.big[rounded].push_back(p);
        d.cached += rounded;
        return true;
    }
    // largest blocks first until at most `keep` bytes are left
    void trim_locked(PerDevice &d, size_t keep) {
        for (auto it = d.big.rbegin(); it != d.big.rend() && d.cached > keep; ++it)
            while (!it->second.empty() && d.cached > keep) {
                (void)hipFree(it->second.back());
                it->second.pop_back();
                d.cached -= it->first;
            }
    }
    void free_all(int device) {   // (an allocation failed: what is kept here may be what is missing)
        std::lock_guard<std::mutex> g(m);
        trim_locked(dev[device], 0);
        dev[device].big.clear();
    }
    void engine_created(int device) { std::lock_guard<std::mutex> g(m); dev[device].engines += 1; }
    void engine_destroyed(int device) {   // the last one of the device: only the default amount stays
        std::lock_guard<std::mutex> g(m);
        PerDevice &d = dev[device];
        if (--d.engines <= 0) { d.engines = 0; trim_locked(d, std::min(limit, kDefaultLimit)); }
    }
};

// Per-engine cache of small device blocks.  A plan of one short segment is otherwise dominated by
// hipMalloc / hipFree (the latter synchronises the device): blocks up to 32 MiB are kept by
// power-of-two size class when a plan lets go of them and handed to the next one.  Every user of
// one pool enqueues on the same engine stream, so a recycled block is ordered after its last use.  (The one exception,
// the transfer ring's own stream, synchronises the engine stream before it touches pooled blocks: stage_file.)
struct DevPool {
    static constexpr int kMinShift = 8, kClasses = 18;   // 256 B .. 32 MiB
    static constexpr size_t kMaxCached = (size_t)512 << 20;
    // Large blocks (> 32 MiB: the arrays of a staged file, the scratch of the BAM decoder) are kept too, by size rounded
    // up to a quarter of a power of two, up to kBigLimit in all: on some hosts a hipMalloc / hipFree pair of a few
    // hundred MB takes tens of milliseconds (measured: 60 ms per call on one box of the pool, < 1 ms on others), which a
    // caller that stages file after file -- or the same file again -- would pay every time.
    static constexpr size_t kBigLimit = (size_t)64 << 30;
    std::mutex m;
    std::vector<void *> bins[kClasses];
    size_t cached = 0;
    std::map<size_t, std::vector<void *>> big;
    size_t big_cached = 0;
    int device = 0;
    static int size_class(size_t bytes) {
        int c = 0;
        while (c < kClasses && ((size_t)1 << (kMinShift + c)) < bytes) ++c;
        return c;   // kClasses: too large to pool
    }
    static size_t class_bytes(int c) { return (size_t)1 << (kMinShift + c); }
    static size_t big_round(size_t bytes) {
        const int k = 63 - __builtin_clzll((unsigned long long)bytes);
        const size_t step = (size_t)1 << (k - 2);
        return (bytes + step - 1) / step * step;
    }
    void *take(int c) {
        {
            std::lock_guard<std::mutex> g(m);
            if (!bins[c].empty()) {
                void *p = bins[c].back();
                bins[c].pop_back();
                cached -= class_bytes(c);
                return p;
            }
        }
        return BigReservoir::get().take(device, class_bytes(c));
    }
    bool give(void *p, int c) {
        std::lock_guard<std::mutex> g(m);
        if (cached + class_bytes(c) > kMaxCached) return false;
        bins[c].push_back(p);
        cached += class_bytes(c);
        return true;
    }
    void *big_take(size_t rounded) {
        {
            std::lock_guard<std::mutex> g(m);
            auto it = big.find(rounded);
            if (it != big.end() && !it->second.empty()) {
                void *p = it->second.back();
                it->second.pop_back();
                big_cached -= rounded;
                return p;
            }
        }
        return BigReservoir::get().take(device, rounded);
    }
    bool big_give(void *p, size_t rounded) {
        std::lock_guard<std::mutex> g(m);
        if (big_cached + rounded > kBigLimit) return false;
        big[rounded].push_back(p);
        big_cached += rounded;
        return true;
    }
    // keep_big: the engine is going away with its streams drained -- its blocks go to the process-wide reservoir
    void drain(bool keep_big) {
        std::lock_guard<std::mutex> g(m);
        for (int c = 0; c < kClasses; ++c) {
            for (void *p : bins[c])
                if (!keep_big || !BigReservoir::get().give(device, p, class_bytes(c))) (void)hipFree(p);
            bins[c].clear();
        }
        cached = 0;
        for (auto &kv : big)
            for (void *p : kv.second)
                if (!keep_big || !BigReservoir::get().give(device, p, kv.first)) (void)hipFree(p);
        big.clear();
        big_cached = 0;
        if (!keep_big) BigReservoir::get().free_all(device);
    }
    ~DevPool() { drain(true); }
};

// The pool a DevBuf without one of its own takes its blocks from: set for the duration of an entry point that
// allocates on behalf of an engine (staging, the BAM decoder), so that every buffer made there -- temporaries and the
// arrays a staged file keeps -- is recycled through that engine's pool.  (All users of one pool enqueue on its engine's
// streams in an order the entry points fix with events, so a recycled block is ordered after its last use.)
static thread_local DevPool *tls_pool = nullptr;
struct PoolScope {
    DevPool *prev;
    explicit PoolScope(DevPool *p) : prev(tls_pool) { tls_pool = p; }
    ~PoolScope() { tls_pool = prev; }
};

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    DevPool *pool = nullptr;   // set: blocks come from / return to this pool
    int pool_class = -1;       // size class of the current block, -1: plain hipMalloc, -2: a large block of `big_bytes`
    size_t big_bytes = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) {
            bool kept = false;
            if (pool && pool_class >= 0) kept = pool->give(p, pool_class);
            else if (pool && pool_class == -2) kept = pool->big_give(p, big_bytes);
            if (!kept) (void)hipFree(p);
        }
        p = nullptr;
        cap = 0;
        pool_class = -1;
        big_bytes = 0;
    }
    int reserve(size_t n) {
        if (n <= cap) return PC_OK;
        release();
        if (n == 0) return PC_OK;
        if (!pool) pool = tls_pool;
        if (pool) {
            const int c = DevPool::size_class(n * sizeof(T));
            if (c < DevPool::kClasses) {
                void *q = pool->take(c);
                if (!q && hipMalloc(&q, DevPool::class_bytes(c)) != hipSuccess) {
                    (void)hipGetLastError();
                    pool->drain(false);   // (what the pool and the reservoir hold may be what is missing)
                    HIP_TRY(hipMalloc(&q, DevPool::class_bytes(c)));
                }
                p = (T *)q;
                cap = DevPool::class_bytes(c) / sizeof(T);
                pool_class = c;
                return PC_OK;
            }
            const size_t rounded = DevPool::big_round(n * sizeof(T));
            void *q = pool->big_take(rounded);
            if (!q && hipMalloc(&q, rounded) != hipSuccess) {
                (void)hipGetLastError();
                pool->drain(false);   // (what the pool and the reservoir hold may be what is missing)
                HIP_TRY(hipMalloc(&q, rounded));
            }
            p = (T *)q;
            cap = rounded / sizeof(T);
            pool_class = -2;
            big_bytes = rounded;
            return PC_OK;
        }
        if (hipMalloc((void **)&p, n * sizeof(T)) != hipSuccess) {   // (no pool of its own: the reservoir of the current device may hold what is missing)
            (void)hipGetLastError();
            p = nullptr;
            int dev_now = 0;
            if (hipGetDevice(&dev_now) == hipSuccess) BigReservoir::get().free_all(dev_now);
            HIP_TRY(hipMalloc((void **)&p, n * sizeof(T)));
        }
        cap = n;
        return PC_OK;
    }
    int upload(const T *src, size_t n, hipStream_t s) {
        int rc = reserve(n);
        if (rc != PC_OK) return rc;
        if (n) HIP_TRY(hipMemcpyAsync(p, src, n * sizeof(T), hipMemcpyHostToDevice, s));
        return PC_OK;
    }
    int upload(const std::vector<T> &v, hipStream_t s) { return upload(v.data(), v.size(), s); }
    void swap(DevBuf &o) {
        std::swap(p, o.p); std::swap(cap, o.cap); std::swap(pool, o.pool); std::swap(pool_class, o.pool_class); std::swap(big_bytes, o.big_bytes);
    }
};

// Page-locked host buffer that only grows.
struct PinnedBuf {
    uint8_t *p = nullptr;
    size_t cap = 0;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    int reserve(size_t n) {
        if (n <= cap) return PC_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        const size_t want = std::max<size_t>(n + n / 2, 512 * 1024);   // covers every short read-back (kSmallRead) from the start
        HIP_TRY(hipHostMalloc((void **)&p, want, hipHostMallocDefault));
        cap = want;
        return PC_OK;
    }
};

// A typed window into a block some DevBuf owns (the tables of a plan share one block and one upload).
template <typename T> struct DevView {
    T *p = nullptr;
};

// Caller-owned (pageable) host arrays <-> HBM at the rate of the PCIe link.  hipMemcpy to or from pageable memory pins
// the pages on the fly, on one thread inside the driver: 25 GB/s for memory the runtime has not seen before, whatever
// the number of calling threads or streams (scripts/ubench/upload_probe.hip; 57 GB/s from page-locked memory on the
// same box).  Here a few host threads move pieces of the arrays through a ring of page-locked slots, each piece with a
// DMA of its own: 53 - 55 GB/s.  Up: a thread takes the next piece, waits until its slot's previous piece has landed,
// copies the piece into the slot and queues the DMA.  Down: it queues the DMA into the slot, waits for it and copies the
// piece out.  Pieces are taken in order and there are more slots than threads, so nobody waits on a piece that has not
// been taken.
struct TransferJob { void *dst; const void *src; size_t bytes; };
struct TransferRing {
    static constexpr int kSlots = 12, kThreadsUp = 6, kThreadsDown = 10;   // (down: the destination's pages are often touched for the first time)
    static constexpr size_t kPiece = (size_t)16 << 20;
    uint8_t *slot[kSlots] = {};
    hipEvent_t ev[kSlots] = {};
    hipStream_t stream = nullptr;
    std::mutex busy;   // one transfer at a time per device
    // One ring per device for the life of the process: page-locking its 192 MB costs as much as staging ten million
    // records, and engines come and go (one per BAMGenomeArray).
    static TransferRing &of(int device) {
        static std::mutex m;
        static std::map<int, TransferRing *> *rings = new std::map<int, TransferRing *>;   // (never destroyed: no HIP call during static destruction)
        std::lock_guard<std::mutex> g(m);
        TransferRing *&r = (*rings)[device];
        if (!r) r = new TransferRing;
        return *r;
    }
    // every job has landed when this returns; `down`: dst is host memory, src device memory.
    // Small transfers take the runtime's own path (`always`: the ring whatever the size)
    int run(int device, const std::vector<TransferJob> &jobs, size_t piece, bool always, bool down = false) {
        std::lock_guard<std::mutex> one(busy);
        HIP_TRY(hipSetDevice(device));
        if (!stream) HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        size_t total = 0;
        for (const auto &j : jobs) total += j.bytes;
        if (total == 0) return PC_OK;
        const hipMemcpyKind kind = down ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice;
        if (total < 4 * kPiece && !always) {
            for (const auto &j : jobs)
                if (j.bytes) HIP_TRY(hipMemcpyAsync(j.dst, j.src, j.bytes, kind, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            return PC_OK;
        }
        piece = std::max<size_t>(1, std::min(piece, kPiece));
        std::vector<TransferJob> pieces;
        for (const auto &j : jobs) {
            if (j.bytes == 0) continue;
            // host memory that is page-locked already (hipHostMalloc, hipHostRegister, a pinned torch tensor) needs no ring
            hipPointerAttribute_t attr;
            const void *host_side = down ? j.dst : j.src;
            if (!always && hipPointerGetAttributes(&attr, host_side) == hipSuccess && attr.type == hipMemoryTypeHost) {
                HIP_TRY(hipMemcpyAsync(j.dst, j.src, j.bytes, kind, stream));
                continue;
            }
            (void)hipGetLastError();   // (an ordinary pointer: some runtimes report it as an error)
            for (size_t off = 0; off < j.bytes; off += piece)
                pieces.push_back({(uint8_t *)j.dst + off, (const uint8_t *)j.src + off, std::min(piece, j.bytes - off)});
        }
        const size_t np = pieces.size();
        for (int k = 0; k < kSlots && np > 0; ++k) {
            if (!slot[k]) HIP_TRY(hipHostMalloc((void **)&slot[k], kPiece, hipHostMallocDefault));
            if (!ev[k]) HIP_TRY(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
        }
        std::unique_ptr<std::atomic<uint8_t>[]> done(new std::atomic<uint8_t>[np]);   // up: DMA queued; down: copied out
        for (size_t k = 0; k < np; ++k) done[k].store(0, std::memory_order_relaxed);
        std::atomic<size_t> next{0};
        std::atomic<int> failed{0};
        std::mutex order;   // DMA and event of a piece are queued together
        auto work = [&]() {
            if (hipSetDevice(device) != hipSuccess) failed.store(1);
            for (;;) {
                const size_t p = next.fetch_add(1);
                if (p >= np) return;
                const int k = (int)(p % kSlots);
                if (p >= (size_t)kSlots) {   // the slot's previous piece
                    while (!done[p - kSlots].load(std::memory_order_acquire)) std::this_thread::yield();
                    if (!down && !failed.load() && hipEventSynchronize(ev[k]) != hipSuccess) failed.store(1);
                }
                if (!failed.load()) {
                    if (!down) std::memcpy(slot[k], pieces[p].src, pieces[p].bytes);
                    {
                        std::lock_guard<std::mutex> lk(order);
                        if (hipMemcpyAsync(down ? (void *)slot[k] : pieces[p].dst, down ? pieces[p].src : (const void *)slot[k], pieces[p].bytes, kind, stream) != hipSuccess ||
                            hipEventRecord(ev[k], stream) != hipSuccess)
                            failed.store(1);
                    }
                    if (down && !failed.load()) {
                        if (hipEventSynchronize(ev[k]) != hipSuccess) failed.store(1);
                        else std::memcpy(pieces[p].dst, slot[k], pieces[p].bytes);
                    }
                }
                done[p].store(1, std::memory_order_release);   // (also after a failure: whoever waits for this slot goes on and ends)
            }
        };
        int T = down ? kThreadsDown : kThreadsUp;
        if (const char *env = getenv("PC_STAGE_UPLOAD_THREADS")) T = std::max(1, std::min(kSlots - 1, atoi(env)));
        T = (int)std::min<size_t>((size_t)T, np);
        std::vector<std::thread> th;
        for (int t = 1; t < T; ++t) th.emplace_back(work);
        if (np > 0) work();
        for (auto &x : th) x.join();
        const hipError_t he = hipStreamSynchronize(stream);
        if (failed.load() || he != hipSuccess) { (void)hipGetLastError(); return fail(PC_ERR_HIP, "transfer ring: a copy failed"); }
        return PC_OK;
    }
};

// CPUs this process may actually use: the smaller of the hardware threads, the affinity mask and the
// container's CFS quota (cgroup v2 cpu.max / v1 cpu.cfs_quota_us).  Pools sized beyond the quota only
// burn it in bursts and are then throttled as a whole.
static int usable_cpus() {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min<unsigned>(n, (unsigned)std::max(1, CPU_COUNT(&set)));
    long long quota = -1, period = -1;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        if (fscanf(f, "%31s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else {
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = -1; fclose(g); }
    }
    if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
    return (int)n;
}

static int stage_threads(int64_t n) {
    int t = std::min(usable_cpus(), 32);
    if (const char *env = getenv("PC_STAGE_THREADS")) t = std::max(1, atoi(env));
    return (int)std::max<int64_t>(1, std::min<int64_t>(t, n / (1 << 20) + 1)); // a thread is not worth < 1 M records
}

struct StagedFile {
    int64_t n = 0, nrun = 0, nlong = 0;
    int W = 1;               // max reference span of the records scanned by the window kernels
    int64_t max_span = 1;    // over all records (long ones too)
    DevBuf<uint2> rec;
    DevBuf<uint32_t> blk_off;
    DevBuf<int2> blk;
    DevBuf<int64_t> tid_bounds;
    DevBuf<uint32_t> long_idx;
    DevBuf<int32_t> long_tid;
    DevBuf<int32_t> long_pmax;
    DevBuf<int64_t> long_tid_bounds;
    DevBuf<uint4> long_rec;
    int64_t ngap = 0;
    DevBuf<uint4> gap_rec;
    DevBuf<int4> gap_runs, long_runs; // first two aligned runs of every side-list record
    DevBuf<int64_t> gap_tid_bounds;
    DevBuf<uint32_t> lin_tab, glin_tab, llin_tab, plin_tab;
    DevBuf<int64_t> lin_off;
    std::vector<int64_t> len_hist; // records per aligned length (host), for cheap warn pre-checks
    int len_min = 65536, len_max = -1; // aligned lengths present
    int slen_min = 0, slen_max = 0;    // ... among the records the 4-byte stream carries
    int tlen_min = 0, tlen_max = 0;    // ... among those and the reads of the run stream (range of the LDS entry table)
    DevBuf<uint32_t> stream;           // 4-byte record stream (pc::stream_word), padded with skip words
    // run stream (aligned runs of multi-run reads with L <= kStreamMaxLen, sorted by contig and run start)
    int64_t nrunrec = 0;
    int Wr = 1;                        // longest run it carries
    DevBuf<uint2> run_rec;
    DevBuf<uint32_t> run_recidx;       // record of every run (for pc_update_flags)
    DevBuf<uint32_t> rlin_tab;
    // long-span reads outside the run stream: the point rules' own long list
    int64_t nxlong = 0;
    int Wg = 1;                        // longest span in the gapped-record list
    DevBuf<uint4> xlong_rec;
    DevBuf<int4> xlong_runs;
    DevBuf<uint32_t> xllin_tab, xplin_tab;
    // wide records (aligned length > 65 535 or > 255 aligned runs): their true {length, run count} next to every
    // long-list entry, and by record index (ascending) for the kernels that start from a record
    int64_t nwide = 0;
    DevBuf<uint2> long_wide, xlong_wide, wide_val;
    DevBuf<uint32_t> wide_rec;
    // SAM FLAG word and MAPQ of every record (pc_set_alignment_sam, or straight from the device decoder): what the
    // vectorised FLAG / MAPQ filter reads
    DevBuf<uint16_t> sam_flag;
    DevBuf<uint8_t> sam_mapq;
    bool have_sam = false;
    DevBuf<uint16_t> sam_nh;           // the NH:i tag of every record (0: none): pc_set_alignment_nh, or straight from the device decoder
    bool have_nh = false;
    // center streams (built at the first center-rule count that needs them; dropped when the host-side filters change)
    DevBuf<uint2> cs_ent[3];
    DevBuf<uint32_t> cs_soff[3];
    int64_t cs_n[3] = {-1, -1, -1};    // entries, -1: not built
    int cs_nib[3] = {-1, -1, -1};      // the nibble the entries were trimmed with
    FileView view() const {
        FileView v;
        v.rec = rec.p; v.blk_off = blk_off.p; v.blk = blk.p; v.tid_bounds = tid_bounds.p;
        v.stream = stream.p;
        v.long_idx = long_idx.p; v.long_tid = long_tid.p; v.long_pmax = long_pmax.p;
        v.long_tid_bounds = long_tid_bounds.p; v.long_rec = long_rec.p; v.n = n; v.nlong = nlong;
        v.gap_rec = gap_rec.p; v.gap_tid_bounds = gap_tid_bounds.p; v.ngap = ngap;
        v.gap_runs = gap_runs.p; v.long_runs = long_runs.p;
        v.lin_tab = lin_tab.p; v.glin_tab = glin_tab.p; v.llin_tab = llin_tab.p; v.plin_tab = plin_tab.p; v.lin_off = lin_off.p;
        v.run_rec = run_rec.p; v.rlin_tab = rlin_tab.p; v.nrunrec = nrunrec;
        v.xlong_rec = xlong_rec.p; v.xlong_runs = xlong_runs.p; v.xllin_tab = xllin_tab.p; v.xplin_tab = xplin_tab.p; v.nxlong = nxlong;
        for (int k = 0; k < 3; ++k) {
            v.cs_ent[k] = cs_n[k] >= 0 ? cs_ent[k].p : nullptr; v.cs_soff[k] = cs_n[k] >= 0 ? cs_soff[k].p : nullptr;
            v.cs_total[k] = cs_n[k] >= 0 ? (uint32_t)cs_n[k] : 0u;
            v.cs_indirect[k] = len_max > 255 ? 1u : 0u;   // (reads beyond the 8-bit fields of a stream entry)
        }
        v.long_wide = nwide ? long_wide.p : nullptr; v.xlong_wide = nwide ? xlong_wide.p : nullptr;
        v.wide_rec = wide_rec.p; v.wide_val = wide_val.p; v.nwide = nwide;
        return v;
    }
};

} // namespace

// Tuning and diagnostic knobs (environment).  Read ONCE, at pc_create -- the query path is
// advertised at tens of microseconds per call and does not look at the environment;
// pc_reload_knobs() re-reads them (tests and experiments drive the scheduling paths with them).
struct Knobs {
    int tile_g = 0;            // PC_TILE_G: window size (0: chosen from the LDS budget)
    int64_t work_r = 49152;    // PC_WORK_R: records per work item (192 KiB of the 4-byte stream)
    int64_t pile = 0;          // PC_PILE: records of a 128-nt sub-window beyond which it is merged through the histogram (0: 12 R)
    int no_small = 0;          // PC_NO_SMALL: no single-wave class for sparse windows
    int small_rows = 0;        // PC_SMALL_ROWS: 1 = multi-row plans (stratified rule) may use the single-wave class too
    int64_t ranges_cg16_max = (int64_t)1 << 19;   // PC_RANGES_CG16_MAX: k_tile_ranges gives a window sixteen lanes while windows x 16 stays within this many threads
    int ranges_cg1 = 0;        // PC_RANGES_CG1: one thread per window in k_tile_ranges whatever the plan's size (tests compare the two forms)
    int64_t first_sync_spare = 65536;   // PC_FIRST_SYNC_SPARE: spare work-list slots from which the first count of a plan reads its item counts back before it launches
    int hist_memset = 0;       // PC_HIST_MEMSET: the compact histogram of a large plan is cleared as a whole before its first count (round 5) instead of slice by slice
    int64_t hist_lazy_bytes = (int64_t)64 << 20;   // PC_HIST_LAZY_BYTES: size from which it is cleared slice by slice (tests: 1)
    int no_stream_probe = 0;   // PC_NO_STREAM_PROBE: keep the engine's streams as created (see settle_streams)
    int no_single = 0;         // PC_NO_SINGLE: one-window plans go through the work lists like any other (tests compare the two paths)
    int plan_build = 0;        // PC_PLAN_BUILD=host|gpu: where pc_plan_create builds the tables (default: on the GPU from 8 192 segments)
    int small_g = 512;         // PC_SMALL_G: queried span a single-wave window may have
    int64_t small_n = 8192;    // PC_SMALL_N: records a single-wave window may scan (C4: 1.25 ms at 2048, 1.22 at 8192, 1.21 at 32768)
    int debug_work = 0;        // PC_DEBUG_WORK: print the queued work items per class (stderr; synchronises)
    int test_stale_counts = 0; // PC_TEST_STALE_COUNTS: pretend the cached work counts are one light item short (exercises the exact-grid guard)
    int center_t1 = 8;         // PC_CENTER_T1 / PC_CENTER_T2: center chunks with more than T1 x (T1*T2 x) the mean candidate
    int center_t2 = 4;         //   count are cut into 4 (8) sub-chunks
    int64_t center_floor = 32768; // PC_CENTER_FLOOR: stream entries below which a chunk is never cut (a wave alone replays ~50 k per ms)
    int center_lds = 0;        // PC_CENTER_LDS: bytes of (unused) LDS per k_center workgroup -- an occupancy throttle for experiments
    int center_per_wave = 0;   // (reserved)
    int center_debug = 0;      // PC_CENTER_DEBUG: wall-clock span of every dispatched wave of k_center, printed after the launch (synchronises)
    void load() {
        *this = Knobs();
        if (const char *env = getenv("PC_TILE_G")) tile_g = std::max(256, atoi(env) / 256 * 256);
        if (const char *env = getenv("PC_WORK_R")) work_r = std::max(1024, atoi(env));
        if (const char *env = getenv("PC_PILE")) pile = std::max<int64_t>(work_r, atoll(env));
        no_small = getenv("PC_NO_SMALL") ? 1 : 0;
        if (const char *env = getenv("PC_SMALL_ROWS")) small_rows = atoi(env);
        no_single = getenv("PC_NO_SINGLE") ? 1 : 0;
        no_stream_probe = getenv("PC_NO_STREAM_PROBE") ? 1 : 0;
        hist_memset = getenv("PC_HIST_MEMSET") ? 1 : 0;
        hist_lazy_bytes = getenv("PC_HIST_LAZY_BYTES") ? std::max<int64_t>(1, atoll(getenv("PC_HIST_LAZY_BYTES"))) : ((int64_t)64 << 20);
        ranges_cg1 = getenv("PC_RANGES_CG1") ? 1 : 0;
        ranges_cg16_max = getenv("PC_RANGES_CG16_MAX") ? atoll(getenv("PC_RANGES_CG16_MAX")) : ((int64_t)1 << 19);
        first_sync_spare = getenv("PC_FIRST_SYNC_SPARE") ? atoll(getenv("PC_FIRST_SYNC_SPARE")) : 65536;
        if (const char *env = getenv("PC_PLAN_BUILD")) plan_build = std::strcmp(env, "host") == 0 ? 1 : (std::strcmp(env, "gpu") == 0 ? 2 : 0);
        if (const char *env = getenv("PC_SMALL_G")) small_g = std::max(64, atoi(env) / 64 * 64);
        if (const char *env = getenv("PC_SMALL_N")) small_n = std::max(64, atoi(env));
        debug_work = getenv("PC_DEBUG_WORK") ? 1 : 0;
        test_stale_counts = getenv("PC_TEST_STALE_COUNTS") ? 1 : 0;
        if (const char *env = getenv("PC_CENTER_T1")) center_t1 = std::max(8, atoi(env));
        if (const char *env = getenv("PC_CENTER_T2")) center_t2 = std::max(1, atoi(env));
        if (const char *env = getenv("PC_CENTER_FLOOR")) center_floor = std::max(64, atoi(env));
        center_debug = getenv("PC_CENTER_DEBUG") ? 1 : 0;
        if (const char *env = getenv("PC_CENTER_LDS")) center_lds = std::max(0, atoi(env));
    }
};

struct pc_engine {
    DevPool pool;   // first member: outlives every buffer of the engine
    DevBuf<uint8_t> plan_scratch[3];   // working arrays of the GPU plan builder (per segment, per piece, per output piece): grown, never shrunk
    Knobs knobs;
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;   // the single-wave kernel of sparse windows runs beside the main one
    static constexpr int kAux = 3;
    hipStream_t aux_stream[kAux] = {nullptr, nullptr, nullptr};   // the upload pieces of the BAM decoder are inflated on these and the main stream in turn, so that one launch fills the tail of the launches before it
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::vector<hipStream_t> parked_streams;   // streams settle_streams traded away (they shared a hardware queue with the main one)
    hipEvent_t ev_pinned = nullptr;      // end of the last copy out of `pinned`
    hipEvent_t ev[8] = {};
    std::vector<StagedFile *> files;
    int ntid = 0;
    DevBuf<FileView> d_files;
    bool files_dirty = true;
    // mapping
    bool have_map = false;
    int kind = PC_MAP_CENTER, param = 0, min_len = 25, max_len = 35, rows = 1, table_len = 0;
    std::vector<int32_t> h_fw, h_rc;
    DevBuf<int32_t> d_fw, d_rc;
    int filt_on = 0, filt_min = 0, filt_max = -1;
    int norm_on = 0;
    double norm_sum = 1.0;
    DevBuf<double> d_inv; // 1.0/m, m = 0..65535 (host-computed IEEE quotients)
    DevBuf<double> d_invh; // the same halved (exact): what the center kernel's fma doubles again
    DevBuf<double> d_cvalh; // [256] half the value of a read by aligned length under the current rule and size filter (k_center_vals)
    // scratch for counting
    DevBuf<uint32_t> d_counters; // [1] unmappable count, [7] sink of the stream probe, [12] exact-grid guard (work counts: pc_plan::d_wcounters)
    DevBuf<uint8_t> d_flags;     // staging buffer of pc_update_flags
    uint8_t *q_host = nullptr;   // pc_query_segment: page-locked buffer the kernel writes the counts of one window into (+ the flag word behind them)
    void *q_dev = nullptr;       //   ... and its address on the device
    uint32_t q_seq = 0;
    bool ff_on = false;          // pc_set_flag_filter: keep (flag & require) == require && (flag & exclude) == 0 && mapq >= min_mapq
    uint32_t ff_require = 0, ff_exclude = 0, ff_min_mapq = 0;
    uint32_t ff_max_nh = 0;      // pc_set_nh_filter: keep only reads with an NH:i tag of at most this many reported alignments (0: no such test)
    bool filter_on() const { return ff_on || ff_max_nh != 0u; }
    bool pinned_busy = false;
    PinnedBuf pinned;            // host side of the plan-table upload (reused: ev_pinned is waited for before it is rewritten)
    PinnedBuf bam_ring[2];       // page-locked halves the image of a large BAM file crosses PCIe through (filled by all host threads)
    hipEvent_t ev_ring[2] = {nullptr, nullptr};
    uint64_t work_generation = 1; // bumped by whatever changes the work list of a plan (alignment files, knobs)
    size_t max_lds = 64 * 1024; // LDS a workgroup may use (160 KiB on gfx950)
    DevBuf<double> d_partial;
    DevBuf<Unmappable> d_unmap;
    double last_ms[6] = {0, 0, 0, 0, 0, 0};
    bool timing_valid = false;
    int prof_level = 0;      // pc_set_profiling: 0 no events, 1 whole call + histogram/center kernel, 2 every phase
    int timed_level = 0;     // level the last pc_count was recorded with
    int64_t last_alg_bytes = 0;
    bool want_center_steps = false;      // pc_center_replay_steps: the next center count reports the replay steps it executed
    int64_t center_steps = 0, center_waves = 0;

    MapParams params() const {
        MapParams mp;
        mp.kind = kind; mp.param = param; mp.min_len = min_len; mp.max_len = max_len; mp.rows = rows;
        mp.filt_on = filt_on; mp.filt_min = filt_min; mp.filt_max = filt_max;
        mp.table_len = table_len; mp.fw = d_fw.p; mp.rc = d_rc.p;
        return mp;
    }
    int W() const {
        int w = 1;
        for (auto *f : files) w = std::max(w, f->W);
        return w;
    }
    int Wg() const {   // halo of the gapped-record list
        int w = 1;
        for (auto *f : files) w = std::max(w, f->Wg);
        return w;
    }
    int Wr() const {   // halo of the run stream: its longest run
        int w = 1;
        for (auto *f : files) w = std::max(w, f->Wr);
        return w;
    }
    int Ws() const {   // halo of the 4-byte record stream: the longest aligned length it carries
        int w = 1;
        for (auto *f : files) w = std::max(w, f->slen_max);
        return w;
    }
};

struct pc_plan {
    pc_engine *e = nullptr;
    int64_t nseg = 0, out_elems = 0, covered = 0;
    int rows = 1, G = 4096;
    uint32_t modes = 0;
    int max_slots = 1;
    int64_t npos = 0; // island positions (hist row length)
    PodVec<Tile> tiles;
    PodVec<Piece> pieces;
    PodVec<OutPiece> opieces;
    size_t n_tiles = 0, n_pieces = 0, n_opieces = 0;   // table sizes (a GPU-built plan keeps its tables in HBM only)
    size_t n_cchunks = 0, n_gchunks = 0;               // ... of the center chunk list and of the gather list, once built
    bool gpu_built = false;      // pc_plan_create built the tables on the GPU (large annotations): they live in HBM only
    bool host_inputs = true;
    DevBuf<uint8_t> d_inputs;    // the caller's segment arrays as uploaded (GPU-built plans)
    DevBuf<GatherSeg> d_gsegs_own;
    bool has_sums = false;       // some slices are summed (out_step 0): the output is an accumulator
    bool out_needs_zero = false; // some queried positions lie outside every tile (unknown contig, clipped)
    bool hist_clean = false;     // compact histogram known to be all zero (point-rule invariant; `hist_lazy`: in the slices of the merged windows)
    bool hist_lazy = false;      // the histogram is large: never cleared as a whole, k_clear_split runs behind every k_tile_ranges
    int hist_kind = -1;          // 0: holds uint32 zeros / merged point-rule windows; -1: freshly allocated, not cleared yet
    std::vector<CenterChunk> cchunks;
    std::vector<GatherSeg> gsegs;
    std::vector<GatherChunk> gchunks;
    bool lazy_center = false;    // large plans: cchunks / gchunks (and the upload of gsegs) wait for the first center count or coordinate export
    bool center_ready = false, gather_ready = false;
    DevBuf<uint8_t> d_tables2, d_tables3;   // ... and live in these blocks (chunks; gather list)
    // host copies for warn evaluation
    std::vector<int32_t> h_tid;
    std::vector<int64_t> h_start, h_end;
    std::vector<uint8_t> h_strand;
    DevBuf<uint8_t> d_tables;   // one block: tiles, pieces, output pieces, chunks, segments, per-tile counters, total
    DevView<Tile> d_tiles;
    DevView<Piece> d_pieces;
    DevView<OutPiece> d_opieces;
    DevView<CenterChunk> d_cchunks;
    DevBuf<uint32_t> d_corder;
    DevBuf<uint32_t> d_ccand;   // candidate records per center chunk
    DevBuf<uint32_t> d_rle_cnt; // run-length encoding of the output: heads per workgroup, their scan,
    DevBuf<int64_t> d_rle_base; // [nwg] bases + [1] total
    DevBuf<int64_t> d_rle_starts;
    DevBuf<unsigned long long> d_rle_values;
    int64_t rle_runs = -1;
    DevBuf<u32x4> d_cranges;    // per (chunk, file): entry range of the near window and candidate range of the long-span list
    DevBuf<u32x2> d_crec;       // per (chunk, file): the near window as a record range (sub-chunks narrow it)
    DevBuf<uint32_t> d_crows;   // per (chunk, file): entry ranges of the chunk's four rows of 16 positions (4 x lo, 4 x hi)
    DevBuf<uint32_t> d_ccounts; // [0] heavy, [1] light entries of the dispatch list, [2..3] sum of the candidate counts, [4..5] entries of all rows, [6..7] 4 x replay steps (row fill)
    DevBuf<CenterSlot> d_cslots; // one descriptor per dispatch entry (plans over one alignment file: k_center2)
    // the center pre-passes (ranges, candidate counts, dispatch order) depend on the plan, the staged files and the
    // knobs only -- not on the mapping rule: kept from count to count while the engine's work generation stands
    uint64_t center_generation = 0;
    int center_W = -1;
    bool center_slots = false;             // the dispatch list has been resolved into descriptors (d_cslots)
    int center_nfiles = 0;                 // ... for this many alignment files (one descriptor per entry and file)
    uint32_t *h_center_counts = nullptr;   // page-locked [2]: heavy, light entries of the list (sizes the grids of later counts)
    hipEvent_t ev_center_counts = nullptr;
    bool center_counts_known = false;
    uint32_t center_counts[2] = {0, 0};
    DevView<GatherSeg> d_gsegs;
    DevView<GatherChunk> d_gchunks;
    DevView<uint32_t> d_tile_items;
    bool tile_items_zero = false;
    // Work lists of the point rules.  They depend on the plan's windows, the staged alignments and the halo of the
    // mapping rule only -- not on the counts -- so they belong to the plan and k_tile_ranges runs once per
    // (plan, WorkKey): a repeated count goes straight to the histogram kernels.
    struct WorkKey {
        uint64_t generation = 0;   // engine work_generation (alignments, host-side filters, knobs)
        int nfiles = 0, G = 0, Wg = 0, Ws = 0, Wr = 0, small_g = 0;
        int64_t R = 0, pile = 0, small_n = 0, cap = 0;
        bool operator==(const WorkKey &o) const {
            return generation == o.generation && nfiles == o.nfiles && G == o.G && Wg == o.Wg && Ws == o.Ws && Wr == o.Wr &&
                   small_g == o.small_g && R == o.R && pile == o.pile && small_n == o.small_n && cap == o.cap;
        }
    };
    WorkKey work_key;
    bool work_valid = false;
    DevBuf<WorkItem> d_work, d_work_small;
    DevBuf<FileRange> d_chain, d_chain_small;   // several files: the ranges of files >= 1 of every (joint) work item
    DevView<uint32_t> d_wcounters;              // [0..3] queued heavy, light, small items, long-span candidates
    bool wcounters_zero = false;
    DevView<uint8_t> d_hist; // uint32 or double; inside d_tables when short, else d_hist_own
    DevBuf<uint8_t> d_hist_own;
    DevBuf<uint8_t> d_out;  // int64 or double
    DevBuf<unsigned long long> d_mr_off;   // pc_mapped_reads_batch: mapped reads before every (segment, file) pair
    DevBuf<uint32_t> d_mr_rec;             // ... and their record indices
    int64_t mr_total = -1;
    DevView<uint8_t> d_total;
    int last_dtype = -1;
    bool counted = false;
    // queued work items per class as the last count of this plan left them (a large plan launches exactly
    // that many workgroups next time instead of the whole list capacity)
    uint32_t *h_work_counts = nullptr;   // page-locked [8]: heavy, light, small, (diagnostics), merged windows
    hipEvent_t ev_work_counts = nullptr;
    uint64_t work_counts_generation = 0; // engine work_generation the read-back belongs to (0: none in flight)
    bool work_counts_known = false;      // the read-back has arrived: work_counts holds it
    uint32_t work_counts[3] = {0, 0, 0};
    uint32_t work_merged = 0;            // windows of the lists that are merged through the compact histogram (what k_gather_split lays out)
    bool exact_grid_used = false;        // some count of this plan launched exact grids: its results are read back with the guard word
    bool guard_pending = false;          // the work lists were rebuilt since the guard word was last read (capacity check)
    ~pc_plan() {
        if (ev_work_counts) (void)hipEventDestroy(ev_work_counts);
        if (h_work_counts) (void)hipHostFree(h_work_counts);
        if (ev_center_counts) (void)hipEventDestroy(ev_center_counts);
        if (h_center_counts) (void)hipHostFree(h_center_counts);
    }

    explicit pc_plan(pc_engine *eng) : e(eng) {
        DevPool *pl = &eng->pool;
        d_tables.pool = pl; d_corder.pool = pl;
        d_ccand.pool = pl; d_rle_cnt.pool = pl; d_rle_base.pool = pl; d_rle_starts.pool = pl; d_rle_values.pool = pl;
        d_cranges.pool = pl; d_crec.pool = pl; d_crows.pool = pl; d_ccounts.pool = pl; d_cslots.pool = pl; d_hist_own.pool = pl; d_out.pool = pl; d_tables2.pool = pl; d_tables3.pool = pl;
        d_work.pool = pl; d_work_small.pool = pl; d_chain.pool = pl; d_chain_small.pool = pl;
        d_inputs.pool = pl; d_gsegs_own.pool = pl;
    }
};

namespace {

int mode_of(uint8_t strand) {
    const bool nofilter = strand & PC_STRAND_NOFILTER;
    const int s = strand & 3;
    if (s == PC_STRAND_REV) return nofilter ? 3 : 1;
    if (s == PC_STRAND_FWD) return nofilter ? 2 : 0;
    return 2; // '.' and undefined: all reads, forward rule
}

int refresh_file_views(pc_engine *e) {
    if (!e->files_dirty) return PC_OK;
    std::vector<FileView> v;
    for (auto *f : e->files) v.push_back(f->view());
    int rc = e->d_files.upload(v, e->stream);
    if (rc != PC_OK) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->files_dirty = false;
    return PC_OK;
}

struct StageClock {
    bool on = getenv("PC_STAGE_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[stage] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

// ---- plans built on the GPU (plan_kernels.hip.h): the host copies of the caller's segment arrays, when a host pass needs them
struct PlanInputLayout {   // the caller's seven segment arrays in one block
    size_t at_tid, at_start, at_end, at_strand, at_off, at_step, at_stride, bytes;
    explicit PlanInputLayout(size_t n) {
        size_t b = 0;
        auto place = [&b](size_t k) { const size_t at = b; b += (k + 255) & ~(size_t)255; return at; };
        at_tid = place(n * 4); at_start = place(n * 8); at_end = place(n * 8); at_strand = place(n);
        at_off = place(n * 8); at_step = place(n); at_stride = place(n * 8);
        bytes = b;
    }
};

int fetch_host_inputs(pc_plan *p) {
    if (p->host_inputs) return PC_OK;
    const size_t n = (size_t)p->nseg;
    const PlanInputLayout L(n);
    HIP_TRY(hipStreamSynchronize(p->e->stream));
    p->h_tid.resize(n); p->h_start.resize(n); p->h_end.resize(n); p->h_strand.resize(n);
    if (n) {
        HIP_TRY(hipMemcpy(p->h_tid.data(), p->d_inputs.p + L.at_tid, n * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(p->h_start.data(), p->d_inputs.p + L.at_start, n * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(p->h_end.data(), p->d_inputs.p + L.at_end, n * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(p->h_strand.data(), p->d_inputs.p + L.at_strand, n, hipMemcpyDeviceToHost));
    }
    p->host_inputs = true;
    return PC_OK;
}

struct Bump {   // carves the working arrays of one stage out of one block
    uint8_t *base = nullptr;
    size_t used = 0;
    template <typename T> T *take(size_t n) {
        used = (used + 255) & ~(size_t)255;
        T *q = base ? (T *)(base + used) : nullptr;
        used += n * sizeof(T);
        return q;
    }
};

inline int bits_for(uint64_t v) { int b = 0; while (b < 64 && (v >> b)) ++b; return std::max(b, 1); }

// The tables of a plan, built on the GPU.  `p` arrives with nseg / out_elems / rows set; on PC_OK it has its tables in
// HBM (d_tables and the views into it), the sizes and flags the host builder sets, and no host copies.
int plan_build_gpu(pc_engine *e, pc_plan *p, int64_t nseg, const int32_t *tid, const int64_t *start, const int64_t *end, const uint8_t *strand,
                   const int64_t *out_off, const int8_t *out_step, const int64_t *row_stride, int64_t out_elems, int rows) {
    using namespace pcplan;
    hipStream_t st = e->stream;
    const int ntid = e->ntid;
    const size_t n = (size_t)nseg;
    StageClock pclk;
    // ---- the caller's arrays
    const PlanInputLayout L(n);
    int rc = p->d_inputs.reserve(std::max<size_t>(L.bytes, 256));
    if (rc == PC_OK) rc = p->d_gsegs_own.reserve(std::max<size_t>(n, 1));
    if (rc != PC_OK) return rc;
    uint8_t *di = p->d_inputs.p;
    HIP_TRY(hipMemcpyAsync(di + L.at_tid, tid, n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(di + L.at_start, start, n * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(di + L.at_end, end, n * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(di + L.at_strand, strand, n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(di + L.at_off, out_off, n * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(di + L.at_step, out_step, n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(di + L.at_stride, row_stride, n * 8, hipMemcpyHostToDevice, st));
    SegIn in;
    in.tid = (const int32_t *)(di + L.at_tid); in.start = (const int64_t *)(di + L.at_start); in.end = (const int64_t *)(di + L.at_end);
    in.strand = di + L.at_strand; in.out_off = (const int64_t *)(di + L.at_off); in.out_step = (const int8_t *)(di + L.at_step);
    in.row_stride = (const int64_t *)(di + L.at_stride);
    pclk.lap("plan(gpu): upload");
    // ---- stage A: per segment
    typedef unsigned long long u64;
    size_t cub_a = 0;
    {
        size_t b = 0;
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, (const u64 *)nullptr, (u64 *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n, 0, 64, st); cub_a = std::max(cub_a, b);
        (void)hipcub::DeviceScan::InclusiveScan(nullptr, b, (const u64 *)nullptr, (u64 *)nullptr, GroupMax(), (int)n, st); cub_a = std::max(cub_a, b);
        (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, (const int64_t *)nullptr, (int64_t *)nullptr, (int)n + 1, st); cub_a = std::max(cub_a, b);
        (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n + 1, st); cub_a = std::max(cub_a, b);
    }
    Bump A;
    u64 *keys = nullptr, *keys2 = nullptr, *ge = nullptr, *pm = nullptr, *island_keys = nullptr;
    uint32_t *ends = nullptr, *ends2 = nullptr, *flags = nullptr, *before = nullptr, *npieces = nullptr, *piece_at = nullptr, *nout = nullptr, *out_at = nullptr;
    Island *islands = nullptr;
    int64_t *lens = nullptr, *offs = nullptr;
    Misc *misc = nullptr;
    uint8_t *cub_tmp = nullptr;
    for (int pass = 0; pass < 2; ++pass) {   // (first pass: sizes)
        A.used = 0;
        misc = A.take<Misc>(1);
        keys = A.take<u64>(n); keys2 = A.take<u64>(n); ends = A.take<uint32_t>(n); ends2 = A.take<uint32_t>(n);
        ge = A.take<u64>(n); pm = A.take<u64>(n); flags = A.take<uint32_t>(n + 1); before = A.take<uint32_t>(n + 1);
        islands = A.take<Island>(n); island_keys = A.take<u64>(n); lens = A.take<int64_t>(n + 1); offs = A.take<int64_t>(n + 1);
        npieces = A.take<uint32_t>(n + 1); piece_at = A.take<uint32_t>(n + 1); nout = A.take<uint32_t>(n + 1); out_at = A.take<uint32_t>(n + 1);
        cub_tmp = A.take<uint8_t>(cub_a + 256);
        if (pass == 0) {
            rc = e->plan_scratch[0].reserve(A.used + 256);
            if (rc != PC_OK) return rc;
            A.base = e->plan_scratch[0].p;
        }
    }
    const unsigned gseg = (unsigned)((n + 255) / 256);
    Misc h;
    std::memset(&h, 0, sizeof(h));
    h.first_bad = ~0ull;
    h.max_slots = 1;
    HIP_TRY(hipMemcpyAsync(misc, &h, sizeof(h), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_plan_segs, dim3(gseg), dim3(256), 0, st, in, nseg, ntid, rows, out_elems, p->d_gsegs_own.p, keys, ends, misc);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&h, misc, sizeof(h), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (h.first_bad != ~0ull) {   // the defect of the lowest segment index, as a serial pass reports it
        const long long b = (long long)(h.first_bad >> 8);
        const int kind = (int)(h.first_bad & 0xffu);
        if (kind == 1) return fail(PC_ERR_ARG, "segment %lld: end < start", b);
        if (kind == 2) return fail(PC_ERR_ARG, "segment %lld: out_step must be +1, -1 or 0 (sum)", b);
        const int64_t len = end[b] - start[b], first = out_off[b], last = out_off[b] + (int64_t)out_step[b] * (len - 1);
        const int64_t lo = std::min(first, last), hi = std::max(first, last) + (int64_t)(rows - 1) * row_stride[b];
        return fail(PC_ERR_ARG, "segment %lld: output slice [%lld,%lld] outside buffer of %lld elements", b, (long long)lo, (long long)hi, (long long)out_elems);
    }
    p->modes = h.modes;
    p->covered = (int64_t)h.covered;
    p->has_sums = h.has_sums != 0;
    int nmodes = 0;
    for (int m = 0; m < kModes; ++m) nmodes += (h.modes >> m) & 1;
    if (nmodes == 0) nmodes = 1;
    {   // window size: as the host builder
        int64_t budget = 0;
        const int G = choose_window(rows, nmodes, h.n_iv, h.iv_len, e->knobs.tile_g, &budget);
        if ((rows > 1 ? 2 : 4) * (int64_t)nmodes * rows * G > 150 * 1024) return fail(PC_ERR_ARG, "pc_plan_create: too many rows (%d) for the LDS window", rows);
        p->G = G;
    }
    const int G = p->G;
    const int split_modes = rows > 1 ? 1 : 0;
    const size_t n_iv = (size_t)h.n_iv;
    pclk.lap("plan(gpu): segments");
    if (n_iv) {
        const unsigned giv = (unsigned)((n_iv + 255) / 256);
        size_t b = cub_a;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(cub_tmp, b, keys, keys2, ends, ends2, (int)n, 0, std::min(64, 33 + bits_for((uint64_t)ntid)), st));
        hipLaunchKernelGGL(k_group_ends, dim3(giv), dim3(256), 0, st, keys2, ends2, misc, ge);
        b = cub_a;
        HIP_TRY(hipcub::DeviceScan::InclusiveScan(cub_tmp, b, ge, pm, GroupMax(), (int)n_iv, st));
        hipLaunchKernelGGL(k_island_flags, dim3(giv), dim3(256), 0, st, keys2, pm, misc, flags, (int64_t)n_iv);
        b = cub_a;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(cub_tmp, b, flags, before, (int)n_iv, st));
        hipLaunchKernelGGL(k_island_fill, dim3(giv), dim3(256), 0, st, keys2, pm, flags, before, misc, islands, island_keys);
        hipLaunchKernelGGL(k_island_lens, dim3(giv), dim3(256), 0, st, misc, islands, lens, npieces, G, (int64_t)n_iv);
        b = cub_a;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(cub_tmp, b, lens, offs, (int)n_iv, st));
        b = cub_a;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(cub_tmp, b, npieces, piece_at, (int)n_iv, st));
        hipLaunchKernelGGL(k_island_offsets, dim3(giv), dim3(256), 0, st, misc, islands, offs, lens, piece_at, npieces);
    }
    if (n) {
        hipLaunchKernelGGL(k_seg_island, dim3(gseg), dim3(256), 0, st, in, nseg, p->d_gsegs_own.p, misc, islands, island_keys, nout, G);
        size_t b = cub_a;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(cub_tmp, b, nout, out_at, (int)n, st));
        hipLaunchKernelGGL(k_out_total, dim3(1), dim3(64), 0, st, misc, out_at, nout, nseg);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&h, misc, sizeof(h), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    p->npos = (int64_t)h.npos;
    p->out_needs_zero = h.needs_zero != 0;
    const size_t n_pieces = h.n_pieces, n_op = h.n_opieces;
    if (n_pieces >= 0x7fffffffu || n_op >= 0x7fffffffu) return fail(PC_ERR_ARG, "pc_plan_create: too many window pieces");
    pclk.lap("plan(gpu): islands");
    // ---- stage B: per piece; stage C: per output piece
    size_t cub_b = 0, cub_c = 0;
    {
        size_t b = 0;
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, (const u64 *)nullptr, (u64 *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n_pieces, 0, 64, st); cub_b = std::max(cub_b, b);
        (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n_pieces + 1, st); cub_b = std::max(cub_b, b);
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n_op, 0, 32, st); cub_c = std::max(cub_c, b);
    }
    Bump B, C;
    u64 *pkeys = nullptr, *pkeys2 = nullptr, *tile_keys = nullptr;
    uint32_t *pidx = nullptr, *pidx2 = nullptr, *new_tile = nullptr, *tbefore = nullptr, *otile = nullptr, *otile2 = nullptr, *oidx = nullptr, *oidx2 = nullptr;
    Piece *praw = nullptr, *psorted = nullptr;
    Tile *tiles_tmp = nullptr;
    OutPiece *oraw = nullptr;
    uint8_t *cub_tmp_b = nullptr, *cub_tmp_c = nullptr;
    for (int pass = 0; pass < 2; ++pass) {
        B.used = 0; C.used = 0;
        pkeys = B.take<u64>(n_pieces); pkeys2 = B.take<u64>(n_pieces); pidx = B.take<uint32_t>(n_pieces); pidx2 = B.take<uint32_t>(n_pieces);
        praw = B.take<Piece>(n_pieces); psorted = B.take<Piece>(n_pieces); new_tile = B.take<uint32_t>(n_pieces + 1); tbefore = B.take<uint32_t>(n_pieces + 1);
        tiles_tmp = B.take<Tile>(n_pieces); tile_keys = B.take<u64>(n_pieces); cub_tmp_b = B.take<uint8_t>(cub_b + 256);
        oraw = C.take<OutPiece>(n_op); otile = C.take<uint32_t>(n_op); otile2 = C.take<uint32_t>(n_op); oidx = C.take<uint32_t>(n_op); oidx2 = C.take<uint32_t>(n_op);
        cub_tmp_c = C.take<uint8_t>(cub_c + 256);
        if (pass == 0) {
            rc = e->plan_scratch[1].reserve(B.used + 256);
            if (rc == PC_OK) rc = e->plan_scratch[2].reserve(C.used + 256);
            if (rc != PC_OK) return rc;
            B.base = e->plan_scratch[1].p; C.base = e->plan_scratch[2].p;
        }
    }
    if (n_pieces) {
        const unsigned gp = (unsigned)((n_pieces + 255) / 256);
        hipLaunchKernelGGL(k_pieces_raw, dim3(gp), dim3(256), 0, st, misc, islands, piece_at, G, pkeys, pidx, praw);
        size_t b = cub_b;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(cub_tmp_b, b, pkeys, pkeys2, pidx, pidx2, (int)n_pieces, 0, std::min(64, 37 + bits_for((uint64_t)ntid)), st));
        hipLaunchKernelGGL(k_pieces_sorted, dim3(gp), dim3(256), 0, st, misc, pkeys2, pidx2, praw, psorted, new_tile, split_modes, (int64_t)n_pieces);
        b = cub_b;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(cub_tmp_b, b, new_tile, tbefore, (int)n_pieces, st));
        hipLaunchKernelGGL(k_tile_fill, dim3(gp), dim3(256), 0, st, misc, pkeys2, psorted, new_tile, tbefore, G, split_modes, tiles_tmp, tile_keys);
    }
    if (n_op) {
        hipLaunchKernelGGL(k_out_raw, dim3(gseg), dim3(256), 0, st, in, nseg, p->d_gsegs_own.p, misc, tile_keys, out_at, nout, G, split_modes, oraw, otile, oidx);
        size_t b = cub_c;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(cub_tmp_c, b, otile, otile2, oidx, oidx2, (int)n_op, 0, std::min(32, bits_for((uint64_t)n_pieces)), st));
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&h, misc, sizeof(h), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const size_t n_tiles = h.n_tiles;
    p->max_slots = (int)h.max_slots;
    p->n_tiles = n_tiles; p->n_pieces = n_pieces; p->n_opieces = n_op;
    pclk.lap("plan(gpu): pieces + tiles + output pieces");
    // ---- the plan's block, laid out as the host builder lays it out
    size_t bytes = 0;
    auto place = [&bytes](size_t k) { const size_t at = bytes; bytes += (k + 255) & ~(size_t)255; return at; };
    const size_t at_tiles = place(n_tiles * sizeof(Tile)), at_pieces = place(n_pieces * sizeof(Piece)), at_opieces = place(n_op * sizeof(OutPiece)),
                 at_cchunks = place(0), at_gsegs = place(0), at_gchunks = place(0),
                 at_items = place((n_tiles + 1) * sizeof(uint32_t)), at_wcounters = place(64), at_total = place(64);
    const size_t hist_full = (size_t)p->npos * (size_t)p->rows * sizeof(double);
    const bool hist_here = hist_full > 0 && hist_full <= 64 * 1024;
    const size_t at_hist = hist_here ? place(hist_full) : 0;
    rc = p->d_tables.reserve(bytes);
    if (rc != PC_OK) return rc;
    uint8_t *d = p->d_tables.p;
    if (n_tiles) HIP_TRY(hipMemcpyAsync(d + at_tiles, tiles_tmp, n_tiles * sizeof(Tile), hipMemcpyDeviceToDevice, st));
    if (n_pieces) HIP_TRY(hipMemcpyAsync(d + at_pieces, psorted, n_pieces * sizeof(Piece), hipMemcpyDeviceToDevice, st));
    if (n_op) hipLaunchKernelGGL(k_out_sorted, dim3((unsigned)((n_op + 255) / 256)), dim3(256), 0, st, misc, otile2, oidx2, oraw, (OutPiece *)(d + at_opieces), (Tile *)(d + at_tiles));
    HIP_TRY(hipMemsetAsync(d + at_items, 0, bytes - at_items, st));
    HIP_TRY(hipGetLastError());
    p->d_tiles.p = (Tile *)(d + at_tiles); p->d_pieces.p = (Piece *)(d + at_pieces);
    p->d_opieces.p = (OutPiece *)(d + at_opieces); p->d_cchunks.p = (CenterChunk *)(d + at_cchunks);
    p->d_gsegs.p = (GatherSeg *)(d + at_gsegs); p->d_gchunks.p = (GatherChunk *)(d + at_gchunks);
    p->d_tile_items.p = (uint32_t *)(d + at_items); p->d_total.p = d + at_total;
    p->d_wcounters.p = (uint32_t *)(d + at_wcounters);
    p->tile_items_zero = true;
    p->wcounters_zero = true;
    if (hist_here) { p->d_hist.p = d + at_hist; p->hist_kind = 0; p->hist_clean = true; }
    p->lazy_center = true;
    p->gpu_built = true;
    p->host_inputs = false;
    pclk.lap("plan(gpu): tables");
    return PC_OK;
}

// The center-only tables of a large plan (see pc_plan_create): built on first use.
// The tables of a large plan that only the center rule (64-position chunks) or only the coordinate export (the
// per-segment gather list) reads are built and uploaded when first asked for, each on its own.
int ensure_center_tables(pc_engine *e, pc_plan *p) {
    if (!p->lazy_center || p->center_ready) return PC_OK;
    if (p->gpu_built) {   // from the tables in HBM: chunks per tile, exclusive sum, fill
        using namespace pcplan;
        hipStream_t st = e->stream;
        const size_t ntl = p->n_tiles;
        size_t tb = 0;
        (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)ntl + 1, st);
        Bump A;
        uint32_t *cnt = nullptr, *at = nullptr;
        uint8_t *tmp = nullptr;
        for (int pass = 0; pass < 2; ++pass) {
            A.used = 0;
            cnt = A.take<uint32_t>(ntl + 1); at = A.take<uint32_t>(ntl + 1); tmp = A.take<uint8_t>(tb + 256);
            if (pass == 0) {
                const int rc0 = e->plan_scratch[0].reserve(A.used + 256);
                if (rc0 != PC_OK) return rc0;
                A.base = e->plan_scratch[0].p;
            }
        }
        uint32_t total = 0;
        if (ntl) {
            const unsigned g = (unsigned)((ntl + 255) / 256);
            HIP_TRY(hipMemsetAsync(cnt + ntl, 0, 4, st));
            hipLaunchKernelGGL(k_cchunk_count, dim3(g), dim3(256), 0, st, p->d_tiles.p, p->d_pieces.p, (uint32_t)ntl, cnt);
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, tb, cnt, at, (int)ntl + 1, st));
            HIP_TRY(hipMemcpyAsync(&total, at + ntl, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            const int rc1 = p->d_tables2.reserve(std::max<size_t>((size_t)total * sizeof(CenterChunk), 256));
            if (rc1 != PC_OK) return rc1;
            if (total) hipLaunchKernelGGL(k_cchunk_fill, dim3(g), dim3(256), 0, st, p->d_tiles.p, p->d_pieces.p, (uint32_t)ntl, at, (CenterChunk *)p->d_tables2.p);
            HIP_TRY(hipGetLastError());
        } else {
            const int rc1 = p->d_tables2.reserve(256);
            if (rc1 != PC_OK) return rc1;
        }
        p->d_cchunks.p = (CenterChunk *)p->d_tables2.p;
        p->n_cchunks = total;
        p->center_ready = true;
        return PC_OK;
    }
    const int PT = std::min(usable_cpus(), 32);
    const size_t ntl = p->tiles.size();
    // chunks per tile, in tile / piece order (what the eager path produces piece by piece)
    std::vector<size_t> at(ntl + 1, 0);
    for (size_t t = 0; t < ntl; ++t) {
        size_t n = 0;
        for (uint32_t i = p->tiles[t].piece_begin; i < p->tiles[t].piece_end; ++i) n += (size_t)(p->pieces[i].len + kWave - 1) / kWave;
        at[t + 1] = at[t] + n;
    }
    p->cchunks.resize(at[ntl]);
    parallel_chunks((int64_t)ntl, PT, [&](int, int64_t tb, int64_t te) {
        for (int64_t t = tb; t < te; ++t) {
            size_t k = at[(size_t)t];
            for (uint32_t i = p->tiles[(size_t)t].piece_begin; i < p->tiles[(size_t)t].piece_end; ++i) {
                const Piece &pc_ = p->pieces[i];
                for (int32_t a = 0; a < pc_.len; a += kWave) {
                    CenterChunk c;
                    c.hist_off = pc_.hist_off + a; c.tid = p->tiles[(size_t)t].tid; c.start = pc_.start + a;
                    c.len = std::min<int32_t>(kWave, pc_.len - a); c.mode = pc_.mode;
                    c.op_begin = p->tiles[(size_t)t].op_begin; c.op_end = p->tiles[(size_t)t].op_end;
                    p->cchunks[k++] = c;
                }
            }
        }
    });
    int rc = p->d_tables2.reserve(std::max<size_t>(p->cchunks.size() * sizeof(CenterChunk), 256));
    if (rc != PC_OK) return rc;
    if (!p->cchunks.empty()) HIP_TRY(hipMemcpyAsync(p->d_tables2.p, p->cchunks.data(), p->cchunks.size() * sizeof(CenterChunk), hipMemcpyHostToDevice, e->stream));
    p->d_cchunks.p = (CenterChunk *)p->d_tables2.p;
    p->n_cchunks = p->cchunks.size();
    p->center_ready = true;
    return PC_OK;
}

int ensure_gather_tables(pc_engine *e, pc_plan *p) {
    if (!p->lazy_center || p->gather_ready) return PC_OK;
    if (p->gpu_built) {   // the per-segment records are in HBM already: (segment, chunk) pairs by count, exclusive sum, fill
        using namespace pcplan;
        hipStream_t st = e->stream;
        const size_t n = (size_t)p->nseg;
        size_t tb = 0;
        (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n + 1, st);
        Bump A;
        uint32_t *cnt = nullptr, *at = nullptr;
        uint8_t *tmp = nullptr;
        for (int pass = 0; pass < 2; ++pass) {
            A.used = 0;
            cnt = A.take<uint32_t>(n + 1); at = A.take<uint32_t>(n + 1); tmp = A.take<uint8_t>(tb + 256);
            if (pass == 0) {
                const int rc0 = e->plan_scratch[0].reserve(A.used + 256);
                if (rc0 != PC_OK) return rc0;
                A.base = e->plan_scratch[0].p;
            }
        }
        uint32_t total = 0;
        const unsigned g = (unsigned)((n + 255) / 256);
        HIP_TRY(hipMemsetAsync(cnt + n, 0, 4, st));
        if (n) hipLaunchKernelGGL(k_gchunk_count, dim3(g), dim3(256), 0, st, p->d_gsegs_own.p, p->nseg, cnt);
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, tb, cnt, at, (int)n + 1, st));
        HIP_TRY(hipMemcpyAsync(&total, at + n, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const int rc1 = p->d_tables3.reserve(std::max<size_t>((size_t)total * sizeof(GatherChunk), 256));
        if (rc1 != PC_OK) return rc1;
        if (total) hipLaunchKernelGGL(k_gchunk_fill, dim3(g), dim3(256), 0, st, p->nseg, cnt, at, (GatherChunk *)p->d_tables3.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));   // (the scratch block may be reused by the next builder call)
        p->d_gsegs.p = p->d_gsegs_own.p;
        p->d_gchunks.p = (GatherChunk *)p->d_tables3.p;
        p->n_gchunks = total;
        p->gather_ready = true;
        return PC_OK;
    }
    const int PT = std::min(usable_cpus(), 32);
    std::vector<size_t> gat((size_t)p->nseg + 1, 0);
    for (int64_t s = 0; s < p->nseg; ++s) gat[(size_t)s + 1] = gat[(size_t)s] + (size_t)((p->gsegs[(size_t)s].len + kGatherChunk - 1) / kGatherChunk);
    p->gchunks.resize(gat[(size_t)p->nseg]);
    parallel_chunks(p->nseg, PT, [&](int, int64_t sb, int64_t se) {
        for (int64_t s = sb; s < se; ++s)
            for (size_t c = 0; c < gat[(size_t)s + 1] - gat[(size_t)s]; ++c) p->gchunks[gat[(size_t)s] + c] = {(uint32_t)s, (uint32_t)c};
    });
    size_t bytes = 0;
    auto place = [&bytes](size_t n) { const size_t a = bytes; bytes += (n + 255) & ~(size_t)255; return a; };
    const size_t at_s = place(p->gsegs.size() * sizeof(GatherSeg)), at_g = place(p->gchunks.size() * sizeof(GatherChunk));
    int rc = p->d_tables3.reserve(std::max<size_t>(bytes, 256));
    if (rc != PC_OK) return rc;
    uint8_t *d = p->d_tables3.p;
    if (!p->gsegs.empty()) HIP_TRY(hipMemcpyAsync(d + at_s, p->gsegs.data(), p->gsegs.size() * sizeof(GatherSeg), hipMemcpyHostToDevice, e->stream));
    if (!p->gchunks.empty()) HIP_TRY(hipMemcpyAsync(d + at_g, p->gchunks.data(), p->gchunks.size() * sizeof(GatherChunk), hipMemcpyHostToDevice, e->stream));
    p->d_gsegs.p = (GatherSeg *)(d + at_s); p->d_gchunks.p = (GatherChunk *)(d + at_g);
    p->n_gchunks = p->gchunks.size();
    p->gather_ready = true;
    return PC_OK;
}

// Center stream `sel` (0 forward reads, 1 reverse reads, 2 all reads) of one staged file: entries per record,
// exclusive sum, scatter -- three passes over the 8-byte records in HBM (see k_center in pc_kernels.hip.h).
int build_center_stream(pc_engine *e, StagedFile *sf, int sel, int nib) {
    hipStream_t st = e->stream;
    const int64_t n = sf->n;
    if (sf->cs_n[sel] >= 0 && sf->cs_nib[sel] == nib) return PC_OK;
    // (one entry per aligned run, counted and scanned in 32 bits: n records + the extra runs of the multi-run ones bounds them)
    if (n + sf->nrun >= (int64_t)0xffffffffu)
        return fail(PC_ERR_ARG, "pc_count: the center rule takes at most 2^32-2 aligned runs per file (%lld records, %lld runs of multi-run reads); split the file",
                    (long long)n, (long long)sf->nrun);
    int rc = PC_OK;
    if (sf->cs_n[sel] < 0) {   // entries per record and their exclusive sum: independent of the nibble
        rc = sf->cs_soff[sel].reserve((size_t)n + 1);
        if (rc != PC_OK) return rc;
        hipLaunchKernelGGL(k_cs_count, dim3((unsigned)((n + 1 + kWG - 1) / kWG)), dim3(kWG), 0, st, sf->rec.p, n, sel, sf->cs_soff[sel].p);
        {
            size_t tmp_bytes = 0;
            DevBuf<uint8_t> d_tmp;
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, sf->cs_soff[sel].p, sf->cs_soff[sel].p, (int)(n + 1), st));
            rc = d_tmp.reserve(tmp_bytes);
            if (rc != PC_OK) return rc;
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tmp_bytes, sf->cs_soff[sel].p, sf->cs_soff[sel].p, (int)(n + 1), st));
            HIP_TRY(hipStreamSynchronize(st));   // d_tmp goes out of scope
        }
        uint32_t total = 0;
        HIP_TRY(hipMemcpy(&total, sf->cs_soff[sel].p + n, sizeof(total), hipMemcpyDeviceToHost));
        rc = sf->cs_ent[sel].reserve((size_t)total + 64);
        if (rc != PC_OK) return rc;
        sf->cs_n[sel] = (int64_t)total;
        e->files_dirty = true;
    }
    hipLaunchKernelGGL(k_cs_scatter, dim3((unsigned)((n + 64 + kWG - 1) / kWG)), dim3(kWG), 0, st, sf->rec.p, sf->blk_off.p, sf->blk.p, n,
                       sel, nib, sf->cs_soff[sel].p, sf->cs_ent[sel].p);
    HIP_TRY(hipGetLastError());
    sf->cs_nib[sel] = nib;
    return PC_OK;
}

// ------------------------------------------------------------------ counting

} // namespace

// ---- which of the engine's streams really run beside the main one
// The runtime multiplexes a process's streams onto a few hardware queues (four by default), and two streams that share
// one run their kernels one after the other.  Two places of the engine count on kernels running side by side -- the
// sparse-window class of a point-rule count (`side_stream`) and the inflate launches of the BAM decoder, which alternate
// between the main stream and `aux_stream[0]` so that one launch fills the tail of the other -- and which queue a new
// stream lands on depends on every stream the process already holds (other engines, torch, RCCL).  Measured in
// bench.py's process, where the headline engine is alive beside the one that decodes the BAM: the two inflate streams
// shared a queue, 139 ms for the 2.9 GB file against 94.5 ms in a process of its own (and 94.5 with GPU_MAX_HW_QUEUES=8).
// So the engine asks: a kernel that waits (at most half a millisecond) for a flag on the main stream, a kernel that sets
// it on the candidate -- seen means the two ran at once.  Candidates that did not are replaced by new streams (the
// runtime hands those to its least-used queue) a few times over; what cannot be had stays as it is: correct, serial.
namespace {
__global__ void k_stream_probe_wait(uint32_t *flag, unsigned long long budget) {
    const unsigned long long t0 = wall_clock64();
    uint32_t seen = 0;
    while ((seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u && wall_clock64() - t0 < budget)
        __builtin_amdgcn_s_sleep(16);
    flag[1] = seen;
}
__global__ void k_stream_probe_set(uint32_t *flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// 1: a kernel on `b` ran while one on `a` was running; 0: it did not; < 0: error
int streams_run_side_by_side(pc_engine *e, hipStream_t a, hipStream_t b, uint32_t *flag) {
    HIP_TRY(hipMemsetAsync(flag, 0, 2 * sizeof(uint32_t), a));
    HIP_TRY(hipEventRecord(e->ev_fork, a));
    HIP_TRY(hipStreamWaitEvent(b, e->ev_fork, 0));
    hipLaunchKernelGGL(k_stream_probe_wait, dim3(1), dim3(1), 0, a, flag, 50000ull);   // wall_clock64 ticks at 100 MHz: 0.5 ms
    hipLaunchKernelGGL(k_stream_probe_set, dim3(1), dim3(1), 0, b, flag);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(a));
    HIP_TRY(hipStreamSynchronize(b));
    uint32_t seen[2] = {0, 0};
    HIP_TRY(hipMemcpy(seen, flag, sizeof(seen), hipMemcpyDeviceToHost));
    return seen[1] ? 1 : 0;
}

// `side_stream` is made a stream that runs beside `stream`, and `aux_stream[0]` one that runs beside both (the BAM
// decoder uploads on the side stream while the main and the auxiliary one inflate), if the process can have such
int settle_streams(pc_engine *e) {
    if (e->knobs.no_stream_probe) return PC_OK;
    DevBuf<uint32_t> flag;
    flag.pool = &e->pool;
    int rc = flag.reserve(2);
    if (rc != PC_OK) return rc;
    auto settle = [&](hipStream_t &cand, std::initializer_list<hipStream_t> beside) -> int {
        for (int attempt = 0;; ++attempt) {
            int ok = 1;
            for (hipStream_t other : beside) {
                ok = streams_run_side_by_side(e, other, cand, flag.p);
                if (ok <= 0) break;
            }
            if (ok < 0) return ok;
            if (ok || attempt == 7) return PC_OK;          // (attempt 7: stays as it is -- correct, serial)
            hipStream_t fresh = nullptr;
            HIP_TRY(hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking));
            e->parked_streams.push_back(cand);             // (destroyed with the engine: destroying it now would hand its queue slot straight back)
            cand = fresh;
        }
    };
    rc = settle(e->side_stream, {e->stream});
    if (rc == PC_OK) rc = settle(e->aux_stream[0], {e->stream, e->side_stream});
    return rc;
}
} // namespace

extern "C" {

const char *pc_last_error(void) { return g_err.c_str(); }
int pc_abi_version(void) { return PC_ABI_VERSION; }

int pc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pc_create(int device, pc_engine **out) {
    if (!out) return fail(PC_ERR_ARG, "pc_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t err = hipGetDeviceCount(&ndev);
    if (err != hipSuccess || ndev <= 0)
        return fail(PC_ERR_HIP, "pc_create: no HIP device available (%s); this engine has no CPU fallback",
                    err == hipSuccess ? "device count is 0" : hipGetErrorString(err));
    if (device < 0 || device >= ndev) return fail(PC_ERR_ARG, "pc_create: device %d out of range [0,%d)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    pc_engine *e = new pc_engine();
    e->device = device;
    e->pool.device = device;
    e->knobs.load();
    BigReservoir::get().engine_created(device);
    HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&e->side_stream, hipStreamNonBlocking));
    for (auto &a : e->aux_stream) HIP_TRY(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&e->ev_pinned, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    {
        int lds_attr = 0;
        if (hipDeviceGetAttribute(&lds_attr, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && lds_attr > 0)
            e->max_lds = (size_t)lds_attr;
    }
    for (auto &ev : e->ev) HIP_TRY(hipEventCreate(&ev));
    std::vector<double> inv(65536);
    inv[0] = 0.0;
    for (int m = 1; m < 65536; ++m) inv[m] = 1.0 / (double)m; // the reference's `1.0 / map_length`
    int rc = e->d_inv.upload(inv, e->stream);
    std::vector<double> invh(65536);
    for (int m = 0; m < 65536; ++m) invh[m] = inv[m] * 0.5;   // exact: a power-of-two scaling of a normal number
    if (rc == PC_OK) rc = e->d_invh.upload(invh, e->stream);
    if (rc == PC_OK) rc = e->d_counters.reserve(16);
    if (rc == PC_OK && hipMemsetAsync(e->d_counters.p, 0, 16 * sizeof(uint32_t), e->stream) != hipSuccess) rc = fail(PC_ERR_HIP, "pc_create: memset failed");
    if (rc == PC_OK && hipStreamSynchronize(e->stream) != hipSuccess) rc = fail(PC_ERR_HIP, "pc_create: sync failed");
    if (rc == PC_OK) rc = settle_streams(e);
    if (rc != PC_OK) {
        pc_destroy(e);
        return rc;
    }
    *out = e;
    return PC_OK;
}

int pc_destroy(pc_engine *e) {
    if (!e) return PC_OK;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (auto *f : e->files) delete f;
    e->files.clear();
    for (auto &ev : e->ev)
        if (ev) (void)hipEventDestroy(ev);
    if (e->side_stream) { (void)hipStreamSynchronize(e->side_stream); (void)hipStreamDestroy(e->side_stream); }
    for (auto &a : e->aux_stream) if (a) { (void)hipStreamSynchronize(a); (void)hipStreamDestroy(a); }
    for (auto a : e->parked_streams) if (a) (void)hipStreamDestroy(a);
    if (e->q_host) (void)hipHostFree(e->q_host);
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_pinned) (void)hipEventDestroy(e->ev_pinned);
    for (auto &x : e->ev_ring) if (x) (void)hipEventDestroy(x);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    const int device = e->device;
    delete e;                                          // (its pool hands its idle blocks to the reservoir)
    BigReservoir::get().engine_destroyed(device);      // the last engine of the device: the reservoir shrinks to its default size
    return PC_OK;
}

int pc_release_cached_memory(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(PC_ERR_ARG, "pc_release_cached_memory: device %d out of range", device);
    HIP_TRY(hipSetDevice(device));
    BigReservoir::get().free_all(device);
    return PC_OK;
}

int pc_host_alloc(pc_engine *e, uint64_t bytes, void **out) {
    if (!e || !out) return fail(PC_ERR_ARG, "pc_host_alloc: engine / out is NULL");
    *out = nullptr;
    HIP_TRY(hipSetDevice(e->device));
    void *p = nullptr;
    HIP_TRY(hipHostMalloc(&p, std::max<uint64_t>(bytes, 8), hipHostMallocDefault));
    *out = p;
    return PC_OK;
}

int pc_host_free(pc_engine *e, void *p) {
    if (!e) return fail(PC_ERR_ARG, "pc_host_free: engine is NULL");
    if (!p) return PC_OK;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));   // (a read-back into it may still be in flight)
    HIP_TRY(hipHostFree(p));
    return PC_OK;
}

int pc_reload_knobs(pc_engine *e) {
    if (!e) return fail(PC_ERR_ARG, "engine is NULL");
    e->knobs.load();
    e->work_generation += 1;
    return PC_OK;
}

int pc_clear_alignments(pc_engine *e) {
    if (!e) return fail(PC_ERR_ARG, "engine is NULL");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    for (auto *f : e->files) delete f;
    e->files.clear();
    e->ntid = 0;
    e->files_dirty = true;
    e->work_generation += 1;
    return PC_OK;
}

int pc_num_files(pc_engine *e) { return e ? (int)e->files.size() : 0; }
int64_t pc_num_records(pc_engine *e, int file) {
    if (!e || file < 0 || file >= (int)e->files.size()) return -1;
    return e->files[file]->n;
}

namespace {
// PC_STAGE_TIMING=1: print where pc_add_alignment_file spends its time (stderr)
} // namespace

int pc_read_records(pc_engine *e, int file, int64_t n, const int64_t *idx, int32_t *tid, int32_t *pos, int32_t *alen, uint8_t *reverse,
                    int32_t *nblk, uint16_t *flag16, uint8_t *mapq) {
    if (!e || file < 0 || file >= (int)e->files.size()) return fail(PC_ERR_ARG, "pc_read_records: bad file index");
    if (n < 0 || (n > 0 && (!idx || !tid || !pos || !alen || !reverse || !nblk))) return fail(PC_ERR_ARG, "pc_read_records: bad arguments");
    if (n == 0) return PC_OK;
    StagedFile *sf = e->files[file];
    for (int64_t k = 0; k < n; ++k)
        if (idx[k] < 0 || idx[k] >= sf->n) return fail(PC_ERR_ARG, "pc_read_records: record index %lld out of range", (long long)idx[k]);
    HIP_TRY(hipSetDevice(e->device));
    PoolScope pool_scope(&e->pool);
    hipStream_t st = e->stream;
    DevBuf<int64_t> d_idx;
    DevBuf<uint8_t> d_out;   // tid, pos, alen, nblk (int32 each), flag16 (u16), reverse, mapq (u8): 20 bytes per record
    int rc = d_idx.upload(idx, (size_t)n, st);
    if (rc == PC_OK) rc = d_out.reserve((size_t)n * 20 + 64);
    if (rc != PC_OK) return rc;
    int32_t *o_tid = (int32_t *)d_out.p, *o_pos = o_tid + n, *o_alen = o_pos + n, *o_nblk = o_alen + n;
    uint16_t *o_f16 = (uint16_t *)(o_nblk + n);
    uint8_t *o_rev = (uint8_t *)(o_f16 + n), *o_mq = o_rev + n;
    hipLaunchKernelGGL(k_gather_records, dim3((unsigned)((n + kWG - 1) / kWG)), dim3(kWG), 0, st, sf->view(), e->ntid, d_idx.p, n,
                       sf->have_sam ? sf->sam_flag.p : nullptr, sf->have_sam ? sf->sam_mapq.p : nullptr, o_tid, o_pos, o_alen, o_rev, o_nblk, o_f16, o_mq);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(tid, o_tid, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(pos, o_pos, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(alen, o_alen, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(nblk, o_nblk, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(reverse, o_rev, (size_t)n, hipMemcpyDeviceToHost, st));
    if (flag16) HIP_TRY(hipMemcpyAsync(flag16, o_f16, (size_t)n * 2, hipMemcpyDeviceToHost, st));
    if (mapq) HIP_TRY(hipMemcpyAsync(mapq, o_mq, (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PC_OK;
}

int pc_read_record_runs(pc_engine *e, int file, int64_t n, const int64_t *idx, const int64_t *run_at, int64_t nruns, int32_t *start, int32_t *len) {
    if (!e || file < 0 || file >= (int)e->files.size()) return fail(PC_ERR_ARG, "pc_read_record_runs: bad file index");
    if (n < 0 || nruns < 0 || (n > 0 && (!idx || !run_at)) || (nruns > 0 && (!start || !len))) return fail(PC_ERR_ARG, "pc_read_record_runs: bad arguments");
    if (n == 0 || nruns == 0) return PC_OK;
    StagedFile *sf = e->files[file];
    for (int64_t k = 0; k < n; ++k) {
        if (idx[k] < 0 || idx[k] >= sf->n) return fail(PC_ERR_ARG, "pc_read_record_runs: record index %lld out of range", (long long)idx[k]);
        if (run_at[k] < 0 || run_at[k] > nruns || (k > 0 && run_at[k] < run_at[k - 1])) return fail(PC_ERR_ARG, "pc_read_record_runs: run offsets must ascend within [0, nruns]");
    }
    HIP_TRY(hipSetDevice(e->device));
    PoolScope pool_scope(&e->pool);
    hipStream_t st = e->stream;
    DevBuf<int64_t> d_idx, d_at;
    DevBuf<int32_t> d_runs;
    int rc = d_idx.upload(idx, (size_t)n, st);
    if (rc == PC_OK) rc = d_at.upload(run_at, (size_t)n, st);
    if (rc == PC_OK) rc = d_runs.reserve((size_t)nruns * 2);
    if (rc != PC_OK) return rc;
    // (the caller sized the run arrays from the run counts pc_read_records gave it: a slot beyond them would be a bug there)
    hipLaunchKernelGGL(k_gather_runs, dim3((unsigned)((n + kWG - 1) / kWG)), dim3(kWG), 0, st, sf->view(), d_idx.p, n, d_at.p, d_runs.p, d_runs.p + nruns);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(start, d_runs.p, (size_t)nruns * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(len, d_runs.p + nruns, (size_t)nruns * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PC_OK;
}

int pc_add_alignment_file(pc_engine *e, int64_t n, int32_t ntid, const int32_t *tid, const int32_t *pos,
                          const uint16_t *alen, const uint8_t *flags, const uint8_t *nblk, int64_t nrun,
                          const int32_t *blk_start, const int32_t *blk_len) {
    return pc_add_alignment_file_wide(e, n, ntid, tid, pos, alen, flags, nblk, nrun, blk_start, blk_len, 0, nullptr, nullptr, nullptr);
}

// `dev`: the columns are in HBM already (a BAM file decoded on the GPU, pc_add_alignment_bam): the host pointers of the
// columns and runs are then NULL and nothing is uploaded or validated (the decoder did that); the wide
// side arrays come from the host either way.
static int stage_file(pc_engine *e, int64_t n, int32_t ntid, const int32_t *tid, const int32_t *pos,
                      const uint16_t *alen, const uint8_t *flags, const uint8_t *nblk, int64_t nrun,
                      const int32_t *blk_start, const int32_t *blk_len, int64_t n_wide, const int64_t *wide_idx,
                      const int32_t *wide_alen, const int32_t *wide_nblk, const pcstage::DevCols *dev);

int pc_add_alignment_file_wide(pc_engine *e, int64_t n, int32_t ntid, const int32_t *tid, const int32_t *pos,
                               const uint16_t *alen, const uint8_t *flags, const uint8_t *nblk, int64_t nrun,
                               const int32_t *blk_start, const int32_t *blk_len, int64_t n_wide, const int64_t *wide_idx,
                               const int32_t *wide_alen, const int32_t *wide_nblk) {
    return stage_file(e, n, ntid, tid, pos, alen, flags, nblk, nrun, blk_start, blk_len, n_wide, wide_idx, wide_alen, wide_nblk, nullptr);
}

static int stage_file(pc_engine *e, int64_t n, int32_t ntid, const int32_t *tid, const int32_t *pos,
                      const uint16_t *alen, const uint8_t *flags, const uint8_t *nblk, int64_t nrun,
                      const int32_t *blk_start, const int32_t *blk_len, int64_t n_wide, const int64_t *wide_idx,
                      const int32_t *wide_alen, const int32_t *wide_nblk, const pcstage::DevCols *dev) {
    if (!e) return fail(PC_ERR_ARG, "engine is NULL");
    if (n < 0 || ntid <= 0 || nrun < 0) return fail(PC_ERR_ARG, "pc_add_alignment_file: bad sizes");
    if (!dev && n > 0 && (!tid || !pos || !alen || !flags || !nblk)) return fail(PC_ERR_ARG, "pc_add_alignment_file: NULL array");
    if (!dev && nrun > 0 && (!blk_start || !blk_len)) return fail(PC_ERR_ARG, "pc_add_alignment_file: NULL run array");
    if (n_wide < 0 || (n_wide > 0 && (!wide_idx || !wide_alen || !wide_nblk))) return fail(PC_ERR_ARG, "pc_add_alignment_file: bad wide-record arrays");
    for (int64_t k = 0; k < n_wide && !dev; ++k) {
        const int64_t i = wide_idx[k];
        if (i < 0 || i >= n || (k > 0 && i <= wide_idx[k - 1])) return fail(PC_ERR_ARG, "pc_add_alignment_file: wide_idx must be ascending record indices");
        if (alen[i] != 0xffffu || nblk[i] != 0xffu) return fail(PC_ERR_ARG, "pc_add_alignment_file: record %lld is listed as wide but its alen / nblk are not 65535 / 255", (long long)i);
        if (wide_alen[k] < 0 || wide_nblk[k] < 0 || (wide_nblk[k] == 0) != (wide_alen[k] == 0) || wide_nblk[k] > wide_alen[k])
            return fail(PC_ERR_ARG, "pc_add_alignment_file: record %lld: bad wide alen / nblk", (long long)i);
    }
    if (n >= (int64_t)0x7fffffff || nrun >= (int64_t)0xffffffffu)
        return fail(PC_ERR_ARG, "pc_add_alignment_file: more than 2^31-2 records per file are not supported");
    if (!e->files.empty() && ntid != e->ntid)
        return fail(PC_ERR_ARG, "pc_add_alignment_file: all files must use the same reference list (ntid %d vs %d)", ntid, e->ntid);
    HIP_TRY(hipSetDevice(e->device));
    PoolScope pool_scope(&e->pool);   // the file's arrays and the temporaries of staging are recycled through the engine's pool

    hipStream_t st = e->stream;
    StageClock clk;

    // ---- caller-owned columns: up to HBM as they are.  The host moves bytes and looks at ONE column, the contigs: sorted,
    // that column is ntid + 1 record bounds, so it does not travel -- a streaming comparison finds the bounds (and the first
    // record out of order or out of range) on the worker threads while the other columns cross PCIe through the ring.
    const bool host_cols = dev == nullptr;
    DevBuf<int32_t> d_pos, d_bs, d_bl;
    DevBuf<uint16_t> d_alen;
    DevBuf<uint8_t> d_flags8, d_nblk8;
    DevBuf<uint32_t> d_wr;
    DevBuf<uint2> d_wv;
    pcstage::DevCols hc;
    std::vector<int64_t> tid_bounds((size_t)ntid + 1, 0);
    int64_t n_ok = n;   // records before the first defect of the contig column
    StagedFile *sf = new StagedFile();
    struct Owner { StagedFile *p; ~Owner() { delete p; } } owner{sf};   // (until the file is the engine's)
    sf->n = n;
    sf->nrun = nrun;
    DevBuf<uint32_t> d_run_at;
    if (host_cols) {
        int rc = d_pos.reserve((size_t)n + 1);
        if (rc == PC_OK) rc = d_alen.reserve((size_t)n + 1);
        if (rc == PC_OK) rc = d_flags8.reserve((size_t)n + 1);
        if (rc == PC_OK) rc = d_nblk8.reserve((size_t)n + 1);
        if (rc == PC_OK) rc = d_bs.reserve((size_t)nrun + 1);
        if (rc == PC_OK) rc = d_bl.reserve((size_t)nrun + 1);
        if (rc != PC_OK) return rc;
        // (the ring fills these blocks on ITS stream: a block the pool recycled may still be read or written by work
        // queued on the engine's stream -- a plan buffer that grew inside an asynchronous pc_count -- and every other user of
        // the pool is ordered on that stream; the ring is the exception, so it starts behind everything queued there)
        HIP_TRY(hipStreamSynchronize(st));
        size_t piece = TransferRing::kPiece;
        if (const char *env = getenv("PC_STAGE_SLICE")) piece = (size_t)std::max<int64_t>(1, std::min<int64_t>(atoll(env), (int64_t)(TransferRing::kPiece / 4))) * 4; // test knob: tiny pieces
        std::vector<TransferJob> jobs;   // (what the first kernel reads goes first)
        jobs.push_back({d_nblk8.p, nblk, (size_t)n});
        jobs.push_back({d_alen.p, alen, (size_t)n * 2});
        jobs.push_back({d_pos.p, pos, (size_t)n * 4});
        jobs.push_back({d_flags8.p, flags, (size_t)n});
        jobs.push_back({d_bs.p, blk_start, (size_t)nrun * 4});
        jobs.push_back({d_bl.p, blk_len, (size_t)nrun * 4});
        struct Joined { std::future<int> f; ~Joined() { if (f.valid()) f.wait(); } } up;   // (joined on every way out, before the buffers go)
        clk.lap("column buffers");
        const int devno = e->device;
        up.f = std::async(std::launch::async, [devno, &jobs, piece]() -> int {
            StageClock uclk;
            const int r = TransferRing::of(devno).run(devno, jobs, piece, getenv("PC_STAGE_SLICE") != nullptr);
            uclk.lap("  (upload thread: ring)");
            return r;
        });
        n_ok = scan_contigs(tid, n, ntid, stage_threads(n), tid_bounds);
        clk.lap("contig bounds");
        if (n_wide > 0) {
            std::vector<uint32_t> wr((size_t)n_wide);
            std::vector<uint2> wv((size_t)n_wide);
            for (int64_t k = 0; k < n_wide; ++k) { wr[(size_t)k] = (uint32_t)wide_idx[k]; wv[(size_t)k] = make_uint2((uint32_t)wide_alen[k], (uint32_t)wide_nblk[k]); }
            rc = d_wr.upload(wr, st);
            if (rc == PC_OK) rc = d_wv.upload(wv, st);
            if (rc == PC_OK && hipStreamSynchronize(st) != hipSuccess) rc = fail(PC_ERR_HIP, "stage: uploading the wide records failed");
        }
        // while the columns cross PCIe: the arrays of the staged file whose sizes follow from the record count alone (a
        // billion records: 20 GB of hipMalloc, 0.12 s when the pool has no such blocks -- a third of the call)
        if (rc == PC_OK) rc = sf->blk_off.reserve((size_t)n + 1);
        if (rc == PC_OK) rc = d_run_at.reserve((size_t)n + 1);
        if (rc == PC_OK) rc = sf->rec.reserve((size_t)n + 2);
        if (rc == PC_OK) rc = sf->stream.reserve((size_t)n + 8);
        clk.lap("arrays of the file (beside the upload)");
        const int urc = up.f.get();
        if (rc != PC_OK) return rc;
        if (urc != PC_OK) return fail(urc, "pc_add_alignment_file: uploading the columns failed");
        hc.tid = nullptr; hc.pos = d_pos.p; hc.alen = d_alen.p; hc.flags = d_flags8.p; hc.nblk = d_nblk8.p;
        hc.blk_start = d_bs.p; hc.blk_len = d_bl.p;
        hc.wide_rec = d_wr.p; hc.wide_val = d_wv.p; hc.n_wide = n_wide;
        dev = &hc;
        clk.lap("columns to HBM");
    }
    // the first defect of the contig column, as the message the caller sees (unless a record before it has one of its own)
    auto contig_defect = [&]() -> int {
        const int64_t i = n_ok;
        if (tid[i] < 0 || tid[i] >= ntid) return fail(PC_ERR_ARG, "record %lld: tid %lld out of range", (long long)i, (long long)tid[i]);
        if (pos[i] < 0) return fail(PC_ERR_ARG, "record %lld: negative position", (long long)i);
        return fail(PC_ERR_UNSORTED, "records are not sorted by (tid, pos) at record %lld; alignment files must be coordinate sorted", (long long)i);
    };

    // ---- where the runs of every record sit in the run arrays (blk_off, kept with the file) / go in the run stream: two
    // exclusive sums
    int64_t nrunrec_total = 0;
    {
        using namespace pcstage;
        DevBuf<uint8_t> d_tmp;
        DevBuf<unsigned long long> d_tot;
        int r = sf->blk_off.reserve((size_t)n + 1);
        if (r == PC_OK) r = d_run_at.reserve((size_t)n + 1);
        if (r == PC_OK) r = d_tot.reserve(2);
        size_t tb = 0;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, d_run_at.p, d_run_at.p, (int)n + 1, st));
        if (r == PC_OK) r = d_tmp.reserve(std::max<size_t>(tb, 16));
        if (r != PC_OK) return r;
        HIP_TRY(hipMemsetAsync(d_tot.p, 0, 16, st));
        hipLaunchKernelGGL(k_cols_runs, dim3((unsigned)std::min<int64_t>((n + 256) / 256, 4096)), dim3(256), 0, st, *dev, n, sf->blk_off.p, d_run_at.p, d_tot.p);
        size_t b2 = tb;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, b2, sf->blk_off.p, sf->blk_off.p, (int)n + 1, st));
        b2 = tb;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, b2, d_run_at.p, d_run_at.p, (int)n + 1, st));
        unsigned long long tot[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(tot, d_tot.p, 16, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));   // (d_tmp goes out of scope)
        if ((int64_t)tot[0] != nrun)
            return fail(PC_ERR_ARG, (int64_t)tot[0] > nrun ? "run arrays shorter than sum of nblk" : "run arrays longer than sum of nblk (%lld vs %lld)",
                        (long long)tot[0], (long long)nrun);
        if (tot[1] >= 0x7fffffffull) return fail(PC_ERR_ARG, "pc_add_alignment_file: more than 2^31-2 aligned runs of multi-run reads per file are not supported");
        nrunrec_total = (int64_t)tot[1];
    }
    clk.lap("run layout (GPU)");
    // run-stream records {run start, len | cum << 8 | L << 16 | flags << 24} and the record of every run, in record order
    DevBuf<uint2> d_val_in;
    DevBuf<uint32_t> d_idx_in;
    int rc = sf->rec.reserve((size_t)n + 2);
    if (rc == PC_OK && nrunrec_total > 0) rc = d_val_in.reserve((size_t)nrunrec_total);
    if (rc == PC_OK && nrunrec_total > 0) rc = d_idx_in.reserve((size_t)nrunrec_total);
    if (rc == PC_OK) rc = sf->stream.reserve((size_t)n + 8);
    if (rc != PC_OK) return rc;
    clk.lap("allocations");

    // ---- one pass over the columns: validation (caller-owned columns), the 8-byte records, the run-stream records, the
    // ends, the statistics of the file (span and length histograms, per-contig bounds) -- stage_kernels.hip.h.  What
    // depends on the statistics of the WHOLE file -- the window halo `wcap` (a span quantile) and with it the long-span
    // class of a record and its stream word -- is derived afterwards (k_classify).
    std::vector<int64_t> span_hist(1026, 0), gap_span_hist(1026, 0), wide_span_hist(1026, 0), len_hist(65536, 0), len1_hist(256, 0), tid_end((size_t)ntid, 0);
    std::vector<int32_t> last_pos((size_t)ntid, -1);   // start of the last record of every contig
    int Wr = 1, rmin = 65536, rmax = -1;
    int64_t max_span = 1;
    if (n > 0) {
        using namespace pcstage;
        DevBuf<unsigned long long> d_stats;   // the statistics block, then the error word of the validation
        DevBuf<int32_t> d_last_pos, d_tid_end;
        DevBuf<int64_t> d_bounds;
        rc = d_stats.reserve(kStatWords + 1);
        if (rc == PC_OK) rc = d_last_pos.reserve((size_t)ntid);
        if (rc == PC_OK) rc = d_tid_end.reserve((size_t)ntid);
        if (rc == PC_OK) rc = d_bounds.reserve((size_t)ntid + 1);
        if (rc != PC_OK) return rc;
        std::vector<unsigned long long> hstats((size_t)kStatWords + 1, 0ull);
        hstats[(size_t)kAtMisc + 1] = 65536ull;   // rmin
        hstats[(size_t)kStatWords] = ~0ull;       // no defect
        std::vector<int32_t> htid_end((size_t)ntid);
        const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
        hipError_t he = hipMemcpyAsync(d_stats.p, hstats.data(), hstats.size() * 8, hipMemcpyHostToDevice, st);
        if (he == hipSuccess) he = hipMemsetAsync(d_tid_end.p, 0, (size_t)ntid * 4, st);
        if (he == hipSuccess && host_cols) {
            he = hipMemcpyAsync(d_bounds.p, tid_bounds.data(), ((size_t)ntid + 1) * 8, hipMemcpyHostToDevice, st);
            if (he == hipSuccess)
                hipLaunchKernelGGL((k_cols_pack<true>), dim3(grid), dim3(256), 0, st, *dev, n_ok, sf->blk_off.p, d_run_at.p, sf->rec.p, d_val_in.p, d_idx_in.p, d_tid_end.p,
                                   d_stats.p, d_bounds.p, ntid, d_stats.p + kStatWords);
            // the verdict before anything is derived from the records
            unsigned long long verdict = ~0ull;
            if (he == hipSuccess) he = hipMemcpyAsync(&verdict, d_stats.p + kStatWords, 8, hipMemcpyDeviceToHost, st);
            if (he == hipSuccess) he = hipStreamSynchronize(st);
            if (he == hipSuccess && verdict != ~0ull) {
                const long long i = (long long)(verdict >> 8);
                switch ((int)(verdict & 0xffu)) {
                case kBadNegPos: return fail(PC_ERR_ARG, "record %lld: negative position", i);
                case kBadOrder: return fail(PC_ERR_UNSORTED, "records are not sorted by (tid, pos) at record %lld; alignment files must be coordinate sorted", i);
                case kBadRuns: return fail(PC_ERR_ARG, "record %lld: aligned runs must be non-empty, ascending and non-adjacent", i);
                case kBadFirstRun: return fail(PC_ERR_ARG, "record %lld: first run must start at pos", i);
                case kBadRunSum: return fail(PC_ERR_ARG, "record %lld: run lengths do not sum to alen", i);
                case kBadLenRuns: return fail(PC_ERR_ARG, "record %lld: nblk/alen mismatch", i);
                default: return fail(PC_ERR_ARG, "record %lld: alignment end beyond 2^31-1", i);
                }
            }
            if (he == hipSuccess && n_ok < n) return contig_defect();
            for (int t = 0; t < ntid; ++t)
                if (tid_bounds[(size_t)t + 1] > tid_bounds[(size_t)t]) last_pos[(size_t)t] = pos[tid_bounds[(size_t)t + 1] - 1];
        } else if (he == hipSuccess) {
            hipLaunchKernelGGL(k_cols_bounds, dim3((unsigned)((ntid + 256) / 256)), dim3(256), 0, st, dev->tid, dev->pos, n, ntid, d_bounds.p, d_last_pos.p);
            hipLaunchKernelGGL((k_cols_pack<false>), dim3(grid), dim3(256), 0, st, *dev, n, sf->blk_off.p, d_run_at.p, sf->rec.p, d_val_in.p, d_idx_in.p, d_tid_end.p,
                               d_stats.p, d_bounds.p, ntid, (unsigned long long *)nullptr);
        }
        if (he == hipSuccess) he = hipMemcpyAsync(hstats.data(), d_stats.p, (size_t)kStatWords * 8, hipMemcpyDeviceToHost, st);
        if (he == hipSuccess && !host_cols) he = hipMemcpyAsync(tid_bounds.data(), d_bounds.p, ((size_t)ntid + 1) * 8, hipMemcpyDeviceToHost, st);
        if (he == hipSuccess) he = hipMemcpyAsync(htid_end.data(), d_tid_end.p, (size_t)ntid * 4, hipMemcpyDeviceToHost, st);
        if (he == hipSuccess && !host_cols) he = hipMemcpyAsync(last_pos.data(), d_last_pos.p, (size_t)ntid * 4, hipMemcpyDeviceToHost, st);
        if (he == hipSuccess) he = hipGetLastError();
        if (he == hipSuccess) he = hipStreamSynchronize(st);
        if (he != hipSuccess) return fail(PC_ERR_HIP, "stage: packing the columns failed: %s", hipGetErrorString(he));
        for (int t = 0; t < ntid; ++t)
            tid_end[(size_t)t] = tid_bounds[(size_t)t + 1] > tid_bounds[(size_t)t] ? (int64_t)htid_end[(size_t)t] : 0;
        for (int k = 0; k < kSpanBins; ++k) {
            span_hist[(size_t)k] = (int64_t)hstats[(size_t)(kAtSpan + k)];
            gap_span_hist[(size_t)k] = (int64_t)hstats[(size_t)(kAtGap + k)];
            wide_span_hist[(size_t)k] = (int64_t)hstats[(size_t)(kAtWide + k)];
        }
        for (int k = 0; k < kLenBins; ++k) len_hist[(size_t)k] = (int64_t)hstats[(size_t)(kAtLen + k)];
        for (int k = 0; k < kLen1Bins; ++k) len1_hist[(size_t)k] = (int64_t)hstats[(size_t)(kAtLen1 + k)];
        Wr = std::max(1, (int)hstats[(size_t)kAtMisc + 0]);
        rmin = (int)hstats[(size_t)kAtMisc + 1];
        rmax = rmin >= 65536 ? -1 : (int)hstats[(size_t)kAtMisc + 2];
        max_span = std::max<int64_t>(1, (int64_t)hstats[(size_t)kAtMisc + 3]);
    }
    clk.lap("validate + pack (GPU)");
    // the window halo W: the smallest span bound (>= 64, <= 1024) that covers >= 99.5% of the records; longer
    // (spliced) reads go through the long-read path
    int wcap = 64;
    {
        int64_t cum = 0;
        const int64_t need = n - n / 200;
        int s0 = 0;
        for (; s0 <= 1024; ++s0) {
            cum += span_hist[(size_t)s0];
            if (cum >= need) break;
        }
        wcap = std::max(64, std::min(s0, 1024));
    }
    int W = 1, Wg = 1;   // the longest span inside the halo: of any record, of a record of the gapped-record list
    for (int s0 = 1; s0 <= wcap; ++s0) {
        if (span_hist[(size_t)s0] > wide_span_hist[(size_t)s0]) W = s0;
        if (gap_span_hist[(size_t)s0]) Wg = s0;
    }
    {   // aligned lengths the 4-byte stream carries (single-run records inside the halo), and those of the run stream
        int smin = 65536, smax = -1;
        for (int L = 0; L <= std::min(wcap, kStreamMaxLen); ++L)
            if (len1_hist[(size_t)L]) { smin = std::min(smin, L); smax = std::max(smax, L); }
        const int tmin = std::min(smin, rmin), tmax = std::max(smax, rmax);
        sf->slen_min = smax >= smin ? smin : 0;
        sf->slen_max = smax >= smin ? smax : 0;
        sf->tlen_min = tmax >= tmin ? tmin : 0;
        sf->tlen_max = tmax >= tmin ? tmax : 0;
    }
    sf->W = W;
    sf->Wg = Wg;
    sf->Wr = Wr;
    sf->max_span = max_span;
    sf->len_hist.swap(len_hist);
    for (int L = 0; L < 65536; ++L)
        if (sf->len_hist[(size_t)L]) { sf->len_min = std::min(sf->len_min, L); sf->len_max = std::max(sf->len_max, L); }

    {   // sentinels behind the last record: two excluded headers, eight skip words (whole quads can always be loaded)
        const uint2 tail_rec[2] = {make_uint2(0u, (uint32_t)PC_FLAG_EXCLUDED << 16), make_uint2(0u, (uint32_t)PC_FLAG_EXCLUDED << 16)};
        uint32_t tail_stream[8];
        for (int k = 0; k < 8; ++k) tail_stream[k] = kStreamSkip;
        if (hipMemcpyAsync(sf->rec.p + n, tail_rec, sizeof(tail_rec), hipMemcpyHostToDevice, e->stream) != hipSuccess ||
            hipMemcpyAsync(sf->stream.p + n, tail_stream, sizeof(tail_stream), hipMemcpyHostToDevice, e->stream) != hipSuccess ||
            hipStreamSynchronize(e->stream) != hipSuccess)
            return fail(PC_ERR_HIP, "pc_add_alignment_file: staging the sentinels failed");
    }
    if (n_wide > 0) {   // the wide records by record index: what the per-record kernels look up
        if (host_cols) {   // (they went up with the columns)
            sf->wide_rec.swap(d_wr);
            sf->wide_val.swap(d_wv);
        } else {
            std::vector<uint32_t> wr((size_t)n_wide);
            std::vector<uint2> wv((size_t)n_wide);
            for (int64_t k = 0; k < n_wide; ++k) { wr[(size_t)k] = (uint32_t)wide_idx[k]; wv[(size_t)k] = make_uint2((uint32_t)wide_alen[k], (uint32_t)wide_nblk[k]); }
            rc = sf->wide_rec.upload(wr, e->stream);
            if (rc == PC_OK) rc = sf->wide_val.upload(wv, e->stream);
            if (rc == PC_OK && hipStreamSynchronize(e->stream) != hipSuccess) rc = fail(PC_ERR_HIP, "stage: sync failed");
            if (rc != PC_OK) return rc;
        }
        sf->nwide = n_wide;
    }
    // the aligned runs as {start, length} pairs
    if (nrun > 0) {
        rc = sf->blk.reserve((size_t)nrun);
        if (rc != PC_OK) return rc;
        hipLaunchKernelGGL(k_zip_runs, dim3((unsigned)((nrun + kWG - 1) / kWG)), dim3(kWG), 0, e->stream, dev->blk_start, dev->blk_len, nrun, sf->blk.p);
    }
    if (nrun == 0) sf->blk_off.release();   // (all zero: no record keeps runs in the run arrays)
    // the class and the 4-byte stream word of every record; per workgroup, the members of the three side lists
    const uint32_t nwg = (uint32_t)((n + 255) / 256);
    DevBuf<uint32_t> d_side_at;   // [3 * nwg + 1]: counts, then their exclusive sum
    uint32_t side_total[3] = {0, 0, 0};
    if (n > 0) {
        using namespace pcstage;
        DevBuf<uint8_t> d_tmp;
        const int nside = 3 * (int)nwg + 1;
        rc = d_side_at.reserve((size_t)nside);
        size_t tb = 0;
        hipError_t he = hipcub::DeviceScan::ExclusiveSum(nullptr, tb, d_side_at.p, d_side_at.p, nside, st);
        if (rc == PC_OK && he == hipSuccess) rc = d_tmp.reserve(std::max<size_t>(tb, 16));
        if (rc != PC_OK) return rc;
        uint32_t at[4] = {0, 0, 0, 0};
        if (he == hipSuccess) he = hipMemsetAsync(d_side_at.p + 3 * (size_t)nwg, 0, 4, st);
        if (he == hipSuccess) {
            hipLaunchKernelGGL(pcstage::k_classify, dim3(nwg), dim3(256), 0, st, sf->rec.p, n, sf->blk_off.p, sf->blk.p, wcap, sf->stream.p, d_side_at.p);
            he = hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tb, d_side_at.p, d_side_at.p, nside, st);
        }
        for (int k = 1; k <= 3 && he == hipSuccess; ++k) he = hipMemcpyAsync(&at[k], d_side_at.p + (size_t)k * nwg, 4, hipMemcpyDeviceToHost, st);
        if (he == hipSuccess) he = hipGetLastError();
        if (he == hipSuccess) he = hipStreamSynchronize(st);   // (d_tmp goes out of scope)
        if (he != hipSuccess) return fail(PC_ERR_HIP, "pc_add_alignment_file: deriving the record stream failed: %s", hipGetErrorString(he));
        for (int k = 0; k < 3; ++k) side_total[k] = at[k + 1] - at[k];
    }
    clk.lap("record stream (GPU)");
    const size_t nrunrec = (size_t)nrunrec_total;
    sf->nrunrec = (int64_t)nrunrec;

    // ---- linear-index layout: one table entry per 2^kLinShift-position bucket of each contig, up to the last
    // record start -- and up to the furthest end of a read of the contig (tid_end): a long-span read reaches
    // windows beyond every record start, and the later runs of gapped reads start there
    std::vector<int64_t> lin_off((size_t)ntid + 1, 0);
    for (int t = 0; t < ntid; ++t) {
        const int64_t b = tid_bounds[(size_t)t], en = tid_bounds[(size_t)t + 1];
        int64_t last = en > b ? (int64_t)last_pos[(size_t)t] : -1;
        if (en > b) last = std::max<int64_t>(last, tid_end[(size_t)t] - 1);
        const int64_t nb = last >= 0 ? (last >> kLinShift) + 1 : 0;
        lin_off[(size_t)t + 1] = lin_off[(size_t)t] + nb + 1;
    }
    const size_t nlin = (size_t)lin_off[(size_t)ntid];

    // ---- side lists and linear-index tables, on the GPU (pc_kernels.hip.h, "side lists"): the records are there
    // already; from the host come only the two small per-contig tables
    rc = sf->tid_bounds.upload(tid_bounds, st);
    if (rc == PC_OK) rc = sf->lin_off.upload(lin_off, st);
    const size_t nlong = side_total[0], ngap = side_total[1], nxlong = side_total[2];
    DevBuf<uint32_t> d_gap_idx, d_xlong_idx;
    if (rc == PC_OK && n > 0) {   // members of the three lists, in record order
        rc = sf->long_idx.reserve(nlong);
        if (rc == PC_OK) rc = d_gap_idx.reserve(ngap);
        if (rc == PC_OK) rc = d_xlong_idx.reserve(nxlong);
        if (rc == PC_OK && nlong + ngap + nxlong > 0)
            hipLaunchKernelGGL(pcstage::k_side_select, dim3(nwg), dim3(256), 0, st, sf->rec.p, n, d_side_at.p, nwg, sf->long_idx.p, d_gap_idx.p, d_xlong_idx.p);
    }
    sf->nlong = (int64_t)nlong;
    sf->ngap = (int64_t)ngap;
    sf->nxlong = (int64_t)nxlong;
    DevBuf<int64_t> d_xlong_bounds;
    DevBuf<int32_t> d_xlong_pmax;
    if (rc == PC_OK) {
        // entries, running maxima of the ends, per-contig ranges
        DevBuf<unsigned long long> d_key, d_key_scanned;
        DevBuf<uint8_t> d_tmp;
        rc = sf->long_rec.reserve(nlong);
        if (rc == PC_OK) rc = sf->long_runs.reserve(nlong);
        if (rc == PC_OK) rc = sf->long_tid.reserve(nlong);
        if (rc == PC_OK) rc = sf->long_pmax.reserve(nlong);
        if (rc == PC_OK && n_wide) rc = sf->long_wide.reserve(nlong);
        if (rc == PC_OK) rc = sf->gap_rec.reserve(ngap);
        if (rc == PC_OK) rc = sf->gap_runs.reserve(ngap);
        if (rc == PC_OK) rc = sf->xlong_rec.reserve(nxlong);
        if (rc == PC_OK) rc = sf->xlong_runs.reserve(nxlong);
        if (rc == PC_OK && n_wide) rc = sf->xlong_wide.reserve(nxlong);
        if (rc == PC_OK) rc = d_xlong_pmax.reserve(nxlong);
        if (rc == PC_OK) rc = d_key.reserve(std::max(nlong, nxlong));
        if (rc == PC_OK) rc = d_key_scanned.reserve(std::max(nlong, nxlong));
        if (rc == PC_OK) rc = sf->long_tid_bounds.reserve((size_t)ntid + 1);
        if (rc == PC_OK) rc = sf->gap_tid_bounds.reserve((size_t)ntid + 1);
        if (rc == PC_OK) rc = d_xlong_bounds.reserve((size_t)ntid + 1);
        hipError_t he = hipSuccess;
        if (rc == PC_OK) {
            size_t need = 0;
            he = hipcub::DeviceScan::InclusiveScan(nullptr, need, d_key.p, d_key_scanned.p, hipcub::Max(), (int)std::max(nlong, nxlong), st);
            if (he == hipSuccess && d_tmp.reserve(std::max<size_t>(need, 16)) != PC_OK) he = hipErrorOutOfMemory;
        }
        auto grid_of = [](size_t m) { return dim3((unsigned)((m + kWG - 1) / kWG)); };
        auto running_max = [&](size_t m, int32_t *pmax) {   // d_key -> pmax
            size_t tb = d_tmp.cap;
            if (he == hipSuccess) he = hipcub::DeviceScan::InclusiveScan(d_tmp.p, tb, d_key.p, d_key_scanned.p, hipcub::Max(), (int)m, st);
            if (he == hipSuccess) hipLaunchKernelGGL(k_unpack_pmax, grid_of(m), dim3(kWG), 0, st, d_key_scanned.p, (int64_t)m, pmax);
        };
        if (rc == PC_OK && he == hipSuccess) {
            if (nlong) {
                hipLaunchKernelGGL(k_side_fill, grid_of(nlong), dim3(kWG), 0, st, sf->long_idx.p, (int64_t)nlong, sf->rec.p, sf->blk_off.p, sf->blk.p,
                                   sf->tid_bounds.p, ntid, sf->wide_rec.p, sf->wide_val.p, n_wide, sf->long_rec.p, sf->long_runs.p, sf->long_tid.p,
                                   d_key.p, n_wide ? sf->long_wide.p : nullptr);
                running_max(nlong, sf->long_pmax.p);
            }
            if (ngap)
                hipLaunchKernelGGL(k_side_fill, grid_of(ngap), dim3(kWG), 0, st, d_gap_idx.p, (int64_t)ngap, sf->rec.p, sf->blk_off.p, sf->blk.p,
                                   sf->tid_bounds.p, ntid, sf->wide_rec.p, sf->wide_val.p, n_wide, sf->gap_rec.p, sf->gap_runs.p, (int32_t *)nullptr,
                                   (unsigned long long *)nullptr, (uint2 *)nullptr);
            if (nxlong) {
                hipLaunchKernelGGL(k_side_fill, grid_of(nxlong), dim3(kWG), 0, st, d_xlong_idx.p, (int64_t)nxlong, sf->rec.p, sf->blk_off.p, sf->blk.p,
                                   sf->tid_bounds.p, ntid, sf->wide_rec.p, sf->wide_val.p, n_wide, sf->xlong_rec.p, sf->xlong_runs.p, (int32_t *)nullptr,
                                   d_key.p, n_wide ? sf->xlong_wide.p : nullptr);
                running_max(nxlong, d_xlong_pmax.p);
            }
            const dim3 gt((unsigned)((ntid + 1 + kWG - 1) / kWG));
            hipLaunchKernelGGL(k_list_bounds, gt, dim3(kWG), 0, st, sf->long_idx.p, (int64_t)nlong, sf->tid_bounds.p, ntid, sf->long_tid_bounds.p);
            hipLaunchKernelGGL(k_list_bounds, gt, dim3(kWG), 0, st, d_gap_idx.p, (int64_t)ngap, sf->tid_bounds.p, ntid, sf->gap_tid_bounds.p);
            hipLaunchKernelGGL(k_list_bounds, gt, dim3(kWG), 0, st, d_xlong_idx.p, (int64_t)nxlong, sf->tid_bounds.p, ntid, d_xlong_bounds.p);
        }
        // the linear-index tables: one bisection per table entry
        if (rc == PC_OK) rc = sf->lin_tab.reserve(nlin);
        if (rc == PC_OK) rc = sf->glin_tab.reserve(nlin);
        if (rc == PC_OK) rc = sf->llin_tab.reserve(nlin);
        if (rc == PC_OK) rc = sf->plin_tab.reserve(nlin);
        if (rc == PC_OK && nxlong) rc = sf->xllin_tab.reserve(nlin);
        if (rc == PC_OK && nxlong) rc = sf->xplin_tab.reserve(nlin);
        if (rc == PC_OK && he == hipSuccess && nlin) {
            const dim3 gl = grid_of(nlin);
            hipLaunchKernelGGL((k_lin_table<0>), gl, dim3(kWG), 0, st, (const void *)sf->rec.p, sf->tid_bounds.p, sf->lin_off.p, ntid, (int64_t)nlin, sf->lin_tab.p);
            hipLaunchKernelGGL((k_lin_table<1>), gl, dim3(kWG), 0, st, (const void *)sf->gap_rec.p, sf->gap_tid_bounds.p, sf->lin_off.p, ntid, (int64_t)nlin, sf->glin_tab.p);
            hipLaunchKernelGGL((k_lin_table<1>), gl, dim3(kWG), 0, st, (const void *)sf->long_rec.p, sf->long_tid_bounds.p, sf->lin_off.p, ntid, (int64_t)nlin, sf->llin_tab.p);
            hipLaunchKernelGGL((k_lin_table<2>), gl, dim3(kWG), 0, st, (const void *)sf->long_pmax.p, sf->long_tid_bounds.p, sf->lin_off.p, ntid, (int64_t)nlin, sf->plin_tab.p);
            if (nxlong) {
                hipLaunchKernelGGL((k_lin_table<1>), gl, dim3(kWG), 0, st, (const void *)sf->xlong_rec.p, d_xlong_bounds.p, sf->lin_off.p, ntid, (int64_t)nlin, sf->xllin_tab.p);
                hipLaunchKernelGGL((k_lin_table<2>), gl, dim3(kWG), 0, st, (const void *)d_xlong_pmax.p, d_xlong_bounds.p, sf->lin_off.p, ntid, (int64_t)nlin, sf->xplin_tab.p);
            }
        }
        if (rc == PC_OK && he == hipSuccess) he = hipGetLastError();
        if (rc == PC_OK && he == hipSuccess) he = hipStreamSynchronize(st);   // the temporaries go out of scope
        if (rc == PC_OK && he != hipSuccess) rc = fail(PC_ERR_HIP, "stage: building the side lists failed: %s", hipGetErrorString(he));
    }
    clk.lap("side lists + linear index (GPU)");
    // ---- run stream: sorted by (contig, run start) on the GPU (radix sort of the 64-bit keys, the 8-byte
    // records and their record indices permuted along), then its linear index by one bisection per bucket
    if (rc == PC_OK && nrunrec) {
        DevBuf<unsigned long long> d_key, d_key_sorted;
        DevBuf<uint32_t> d_ord, d_ord_sorted;
        DevBuf<uint8_t> d_tmp;
        rc = d_key.reserve(nrunrec);
        if (rc == PC_OK) rc = d_ord.reserve(nrunrec);
        if (rc == PC_OK) rc = d_key_sorted.reserve(nrunrec);
        if (rc == PC_OK) rc = d_ord_sorted.reserve(nrunrec);
        if (rc == PC_OK) rc = sf->run_rec.reserve(nrunrec + 1);
        if (rc == PC_OK) rc = sf->run_recidx.reserve(nrunrec);
        if (rc == PC_OK) rc = sf->rlin_tab.reserve(nlin);
        if (rc == PC_OK)   // sort keys (contig << 32 | run start) and the identity permutation, made on the GPU
            hipLaunchKernelGGL(k_run_keys, dim3((unsigned)((nrunrec + kWG - 1) / kWG)), dim3(kWG), 0, e->stream, d_val_in.p, d_idx_in.p,
                               (int64_t)nrunrec, sf->tid_bounds.p, ntid, d_key.p, d_ord.p);
        if (rc == PC_OK) {
            size_t tmp_bytes = 0;
            const int end_bit = 32 + (ntid > 1 ? 32 - __builtin_clz((unsigned)(ntid - 1)) : 1);
            hipError_t he = hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_key.p, d_key_sorted.p, d_ord.p, d_ord_sorted.p,
                                                               (int)nrunrec, 0, end_bit, e->stream);
            if (he == hipSuccess) rc = d_tmp.reserve(tmp_bytes);
            if (he == hipSuccess && rc == PC_OK)
                he = hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tmp_bytes, d_key.p, d_key_sorted.p, d_ord.p, d_ord_sorted.p, (int)nrunrec,
                                                        0, end_bit, e->stream);   // stable: equal starts keep record order
            if (he != hipSuccess) rc = fail(PC_ERR_HIP, "stage: sorting the run stream failed: %s", hipGetErrorString(he));
        }
        if (rc == PC_OK) {
            const unsigned grid = (unsigned)((nrunrec + kWG - 1) / kWG);
            hipLaunchKernelGGL(k_run_gather, dim3(grid), dim3(kWG), 0, e->stream, d_ord_sorted.p, d_val_in.p, d_idx_in.p, (int64_t)nrunrec,
                               sf->run_rec.p, sf->run_recidx.p);
            hipLaunchKernelGGL(k_run_lin, dim3((unsigned)((nlin + kWG - 1) / kWG)), dim3(kWG), 0, e->stream, d_key_sorted.p, (int64_t)nrunrec,
                               sf->lin_off.p, ntid, (int64_t)nlin, sf->rlin_tab.p);
            const uint2 tail_run = make_uint2(0u, (uint32_t)PC_FLAG_EXCLUDED << 24);
            if (hipMemcpyAsync(sf->run_rec.p + nrunrec, &tail_run, sizeof(tail_run), hipMemcpyHostToDevice, e->stream) != hipSuccess ||
                hipGetLastError() != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess)
                rc = fail(PC_ERR_HIP, "stage: building the run stream failed");
        }
    }
    if (rc != PC_OK) return rc;
    clk.lap("run stream (GPU sort)");
    owner.p = nullptr;
    e->files.push_back(sf);
    e->ntid = ntid;
    e->files_dirty = true;
    e->work_generation += 1;
    return PC_OK;
}

static int propagate_record_flags(pc_engine *e, StagedFile *sf);

int pc_update_flags(pc_engine *e, int file, int64_t n, const uint8_t *flags) {
    if (!e || file < 0 || file >= (int)e->files.size()) return fail(PC_ERR_ARG, "pc_update_flags: bad file index");
    StagedFile *sf = e->files[file];
    if (n != sf->n || (n > 0 && !flags)) return fail(PC_ERR_ARG, "pc_update_flags: wrong record count");
    HIP_TRY(hipSetDevice(e->device));
    if (n == 0) return PC_OK;
    // the flags go up (1 byte per record); every staged copy of the headers is patched in HBM
    int rc = e->d_flags.reserve((size_t)n);
    if (rc != PC_OK) return rc;
    hipStream_t st = e->stream;
    HIP_TRY(hipMemcpyAsync(e->d_flags.p, flags, (size_t)n, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_update_flags, dim3((unsigned)((n + kWG - 1) / kWG)), dim3(kWG), 0, st, sf->rec.p, sf->stream.p,
                       e->d_flags.p, n);
    if (e->filter_on() && (sf->have_sam || !e->ff_on) && (sf->have_nh || !e->ff_max_nh))   // the FLAG / MAPQ / NH filter's verdicts on top of the caller's
        hipLaunchKernelGGL(k_flag_filter, dim3((unsigned)((n + kWG - 1) / kWG)), dim3(kWG), 0, st, sf->rec.p, sf->stream.p, sf->have_sam ? sf->sam_flag.p : nullptr,
                           sf->have_sam ? sf->sam_mapq.p : nullptr, n, 1u, e->ff_on ? e->ff_require : 0u, e->ff_on ? e->ff_exclude : 0u, e->ff_on ? e->ff_min_mapq : 0u,
                           sf->have_nh ? sf->sam_nh.p : nullptr, e->ff_max_nh);
    const int prc = propagate_record_flags(e, sf);
    if (prc != PC_OK) return prc;
    HIP_TRY(hipStreamSynchronize(st)); // the caller's flag buffer may go away
    return PC_OK;
}

// The strand / excluded bits of the packed records have changed: copy them into every other staged form of the
// headers (side lists, run stream), drop what was derived from them (center streams, work lists).  Asynchronous.
static int propagate_record_flags(pc_engine *e, StagedFile *sf) {
    hipStream_t st = e->stream;
    if (sf->nlong)
        hipLaunchKernelGGL(k_update_side_flags, dim3((unsigned)((sf->nlong + kWG - 1) / kWG)), dim3(kWG), 0, st, sf->long_rec.p,
                           sf->nlong, sf->rec.p);
    if (sf->ngap)
        hipLaunchKernelGGL(k_update_side_flags, dim3((unsigned)((sf->ngap + kWG - 1) / kWG)), dim3(kWG), 0, st, sf->gap_rec.p,
                           sf->ngap, sf->rec.p);
    if (sf->nxlong)
        hipLaunchKernelGGL(k_update_side_flags, dim3((unsigned)((sf->nxlong + kWG - 1) / kWG)), dim3(kWG), 0, st, sf->xlong_rec.p,
                           sf->nxlong, sf->rec.p);
    if (sf->nrunrec)
        hipLaunchKernelGGL(k_update_run_flags, dim3((unsigned)((sf->nrunrec + kWG - 1) / kWG)), dim3(kWG), 0, st, sf->run_rec.p,
                           sf->run_recidx.p, sf->nrunrec, sf->rec.p);
    HIP_TRY(hipGetLastError());
    for (int k = 0; k < 3; ++k) sf->cs_n[k] = -1;   // the center streams leave excluded reads out: rebuilt at the next center count
    e->files_dirty = true;
    e->work_generation += 1;
    return PC_OK;
}

// (re-)apply the engine's FLAG / MAPQ filter to one staged file
static int apply_flag_filter(pc_engine *e, StagedFile *sf) {
    if (sf->n == 0) return PC_OK;
    if (e->ff_on && !sf->have_sam)
        return fail(PC_ERR_STATE, "a FLAG / MAPQ filter is set but an alignment file was staged without its FLAG / MAPQ columns (pc_set_alignment_sam)");
    if (e->ff_max_nh && !sf->have_nh)
        return fail(PC_ERR_STATE, "an NH filter is set but an alignment file was staged without its NH column (pc_set_alignment_nh)");
    if (!sf->have_sam && !sf->have_nh && !e->filter_on()) {
        // nothing to read the verdicts from, and nothing to undo: a file without the columns never had the filter applied
        return PC_OK;
    }
    hipLaunchKernelGGL(k_flag_filter, dim3((unsigned)((sf->n + kWG - 1) / kWG)), dim3(kWG), 0, e->stream, sf->rec.p, sf->stream.p,
                       sf->have_sam ? sf->sam_flag.p : nullptr, sf->have_sam ? sf->sam_mapq.p : nullptr, sf->n, e->filter_on() ? 1u : 0u,
                       e->ff_on ? e->ff_require : 0u, e->ff_on ? e->ff_exclude : 0u, e->ff_on ? e->ff_min_mapq : 0u,
                       sf->have_nh ? sf->sam_nh.p : nullptr, e->ff_max_nh);
    return propagate_record_flags(e, sf);
}

int pc_set_alignment_sam(pc_engine *e, int file, int64_t n, const uint16_t *flag, const uint8_t *mapq) {
    if (!e || file < 0 || file >= (int)e->files.size()) return fail(PC_ERR_ARG, "pc_set_alignment_sam: bad file index");
    StagedFile *sf = e->files[file];
    if (n != sf->n || (n > 0 && (!flag || !mapq))) return fail(PC_ERR_ARG, "pc_set_alignment_sam: wrong record count");
    HIP_TRY(hipSetDevice(e->device));
    if (n > 0) {
        PoolScope pool_scope(&e->pool);
        int rc = sf->sam_flag.reserve((size_t)n);
        if (rc == PC_OK) rc = sf->sam_mapq.reserve((size_t)n);
        if (rc != PC_OK) return rc;
        HIP_TRY(hipMemcpyAsync(sf->sam_flag.p, flag, (size_t)n * 2, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemcpyAsync(sf->sam_mapq.p, mapq, (size_t)n, hipMemcpyHostToDevice, e->stream));
    }
    sf->have_sam = true;
    int rc = PC_OK;
    if (e->ff_on && (sf->have_nh || !e->ff_max_nh)) rc = apply_flag_filter(e, sf);
    HIP_TRY(hipStreamSynchronize(e->stream));   // the caller's arrays may go away
    return rc;
}

int pc_set_alignment_nh(pc_engine *e, int file, int64_t n, const uint16_t *nh) {
    if (!e || file < 0 || file >= (int)e->files.size()) return fail(PC_ERR_ARG, "pc_set_alignment_nh: bad file index");
    StagedFile *sf = e->files[file];
    if (n != sf->n || (n > 0 && !nh)) return fail(PC_ERR_ARG, "pc_set_alignment_nh: wrong record count");
    HIP_TRY(hipSetDevice(e->device));
    if (n > 0) {
        PoolScope pool_scope(&e->pool);
        const int rc = sf->sam_nh.reserve((size_t)n);
        if (rc != PC_OK) return rc;
        HIP_TRY(hipMemcpyAsync(sf->sam_nh.p, nh, (size_t)n * 2, hipMemcpyHostToDevice, e->stream));
    }
    sf->have_nh = true;
    int rc = PC_OK;
    if (e->ff_max_nh && (sf->have_sam || !e->ff_on)) rc = apply_flag_filter(e, sf);
    HIP_TRY(hipStreamSynchronize(e->stream));   // the caller's array may go away
    return rc;
}

int pc_set_nh_filter(pc_engine *e, int max_nh) {
    if (!e) return fail(PC_ERR_ARG, "engine is NULL");
    if (max_nh < 0 || max_nh > 65535) return fail(PC_ERR_ARG, "pc_set_nh_filter: max_nh is 0 (off) .. 65535");
    if (max_nh)
        for (size_t f = 0; f < e->files.size(); ++f)
            if (e->files[f]->n > 0 && !e->files[f]->have_nh)
                return fail(PC_ERR_STATE, "pc_set_nh_filter: alignment file %d was staged without its NH column (pc_set_alignment_nh)", (int)f);
    HIP_TRY(hipSetDevice(e->device));
    if (e->ff_max_nh == (uint32_t)max_nh) return PC_OK;
    e->ff_max_nh = (uint32_t)max_nh;
    for (StagedFile *sf : e->files) {
        if (!sf->have_nh && !sf->have_sam) continue;
        if ((e->ff_on && !sf->have_sam)) continue;   // (pc_count refuses such a file until its columns arrive)
        const int rc = apply_flag_filter(e, sf);
        if (rc != PC_OK) return rc;
    }
    return PC_OK;
}

int pc_set_flag_filter(pc_engine *e, int enabled, uint32_t require, uint32_t exclude, int min_mapq) {
    if (!e) return fail(PC_ERR_ARG, "engine is NULL");
    if (enabled && (require > 0xffffu || exclude > 0xffffu || min_mapq < 0 || min_mapq > 255))
        return fail(PC_ERR_ARG, "pc_set_flag_filter: FLAG masks are 16-bit, MAPQ is 0 .. 255");
    if (enabled)
        for (size_t f = 0; f < e->files.size(); ++f)
            if (e->files[f]->n > 0 && !e->files[f]->have_sam)
                return fail(PC_ERR_STATE, "pc_set_flag_filter: alignment file %d was staged without its FLAG / MAPQ columns (pc_set_alignment_sam)", (int)f);
    HIP_TRY(hipSetDevice(e->device));
    const bool was_on = e->ff_on;
    const bool same = was_on == (enabled != 0) && (!enabled || (e->ff_require == require && e->ff_exclude == exclude && e->ff_min_mapq == (uint32_t)min_mapq));
    if (same) return PC_OK;
    e->ff_on = enabled != 0;
    e->ff_require = enabled ? require : 0u; e->ff_exclude = enabled ? exclude : 0u; e->ff_min_mapq = enabled ? (uint32_t)min_mapq : 0u;
    for (StagedFile *sf : e->files) {
        if (!sf->have_sam) continue;
        if (e->ff_max_nh && !sf->have_nh) continue;   // (pc_count refuses such a file until its NH column arrives)
        const int rc = apply_flag_filter(e, sf);
        if (rc != PC_OK) return rc;
    }
    return PC_OK;
}

int pc_set_mapping(pc_engine *e, int kind, int param, const int32_t *fw, const int32_t *rc, int table_len,
                   int min_len, int max_len) {
    if (!e) return fail(PC_ERR_ARG, "engine is NULL");
    if (kind < PC_MAP_FIVE || kind > PC_MAP_STRAT5) return fail(PC_ERR_ARG, "pc_set_mapping: unknown kind %d", kind);
    if ((kind == PC_MAP_FIVE || kind == PC_MAP_THREE || kind == PC_MAP_CENTER) && param < 0)
        return fail(PC_ERR_ARG, "pc_set_mapping: offset/nibble must be >= 0, got %d", param);
    int rows = 1;
    if (kind == PC_MAP_VAR5 || kind == PC_MAP_STRAT5) {
        if (!fw || !rc || table_len <= 0 || table_len > 65536) return fail(PC_ERR_ARG, "pc_set_mapping: offset tables required");
        for (int L = 0; L < table_len; ++L) {
            if (fw[L] < -1 || rc[L] < -1 || (fw[L] >= 0 && fw[L] >= std::max(L, 1)) || (rc[L] >= 0 && rc[L] >= std::max(L, 1)))
                return fail(PC_ERR_ARG, "pc_set_mapping: table entry for length %d out of range (the reference would index read.positions out of bounds)", L);
            if ((fw[L] < 0) != (rc[L] < 0)) return fail(PC_ERR_ARG, "pc_set_mapping: forward/reverse tables disagree on length %d", L);
        }
    }
    if (kind == PC_MAP_STRAT5) {
        if (max_len <= min_len) return fail(PC_ERR_ARG, "pc_set_mapping: max length must be > min length"); // :716-717
        if (max_len >= table_len) return fail(PC_ERR_ARG, "pc_set_mapping: max length beyond offset table");
        rows = max_len - min_len + 1;
    }
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->kind = kind; e->param = param; e->min_len = min_len; e->max_len = max_len; e->rows = rows;
    e->table_len = 0;
    e->h_fw.clear(); e->h_rc.clear();
    if (kind == PC_MAP_VAR5 || kind == PC_MAP_STRAT5) {
        e->h_fw.assign(fw, fw + table_len);
        e->h_rc.assign(rc, rc + table_len);
        int r = e->d_fw.upload(e->h_fw, e->stream);
        if (r == PC_OK) r = e->d_rc.upload(e->h_rc, e->stream);
        if (r != PC_OK) return r;
        HIP_TRY(hipStreamSynchronize(e->stream));
        e->table_len = table_len;
    }
    e->have_map = true;
    return PC_OK;
}

int pc_set_size_filter(pc_engine *e, int enabled, int min_len, int max_len) {
    if (!e) return fail(PC_ERR_ARG, "engine is NULL");
    if (enabled) {
        if (max_len != -1 && max_len < min_len) return fail(PC_ERR_ARG, "Alignment size filter: max read length must be >= min read length");
        if (min_len < 1) return fail(PC_ERR_ARG, "Alignment size filter: min read length must be >= 1. Got %d", min_len);
    }
    e->filt_on = enabled ? 1 : 0;
    e->filt_min = min_len;
    e->filt_max = max_len;
    return PC_OK;
}

int pc_set_normalize(pc_engine *e, int enabled, double sum) {
    if (!e) return fail(PC_ERR_ARG, "engine is NULL");
    e->norm_on = enabled ? 1 : 0;
    e->norm_sum = sum;
    return PC_OK;
}

int pc_mapping_rows(pc_engine *e) { return e ? e->rows : 0; }

// ------------------------------------------------------------------ plan
int pc_plan_create(pc_engine *e, int64_t nseg, const int32_t *tid, const int64_t *start, const int64_t *end,
                   const uint8_t *strand, const int64_t *out_off, const int8_t *out_step,
                   const int64_t *row_stride, int64_t out_elems, int rows, pc_plan **out) {
    if (!e || !out) return fail(PC_ERR_ARG, "pc_plan_create: NULL argument");
    *out = nullptr;
    if (nseg < 0 || out_elems < 0 || rows < 1) return fail(PC_ERR_ARG, "pc_plan_create: bad sizes");
    if (nseg > 0 && (!tid || !start || !end || !strand || !out_off || !out_step || !row_stride))
        return fail(PC_ERR_ARG, "pc_plan_create: NULL array");
    if (nseg >= (int64_t)0x7fffffff) return fail(PC_ERR_ARG, "pc_plan_create: too many segments");
    HIP_TRY(hipSetDevice(e->device));
    const int ntid = e->ntid;

    // large annotations: every pass of the builder as a kernel, a radix sort or a scan (plan_kernels.hip.h); the host
    // builder below is what small plans -- and PC_PLAN_BUILD=host -- take, and what the GPU tables are tested against
    if (nseg > 0 && (int64_t)ntid < ((int64_t)1 << pcplan::kTidBits) && (e->knobs.plan_build == 2 || (e->knobs.plan_build == 0 && nseg >= (1 << 13)))) {
        pc_plan *gp = new pc_plan(e);
        gp->nseg = nseg;
        gp->out_elems = out_elems;
        gp->rows = rows;
        const int grc = plan_build_gpu(e, gp, nseg, tid, start, end, strand, out_off, out_step, row_stride, out_elems, rows);
        if (grc != PC_OK) {
            (void)hipStreamSynchronize(e->stream);
            delete gp;
            return grc;
        }
        *out = gp;
        return PC_OK;
    }

    StageClock pclk;
    struct Iv { int32_t tid; int32_t mode; int64_t s, e; };
    std::vector<Iv> ivs;
    ivs.reserve((size_t)nseg);
    pc_plan *p = new pc_plan(e);
    p->nseg = nseg;
    p->out_elems = out_elems;
    p->rows = rows;
    p->gsegs.resize((size_t)nseg);
    p->h_tid.assign(tid, tid + nseg);
    p->h_start.assign(start, start + nseg);
    p->h_end.assign(end, end + nseg);
    p->h_strand.assign(strand, strand + nseg);
    const int64_t kMaxPos = 0x7fffffffLL;
    uint32_t modes = 0;
    // (large annotations: the passes over the segments and the per-contig sorts run on a thread pool)
    const int PT = nseg >= (1 << 16) ? std::min(usable_cpus(), 32) : 1;
    {
        struct SegPart { std::vector<Iv> ivs; uint32_t modes = 0; int64_t covered = 0; bool has_sums = false; int64_t bad = -1; int bad_kind = 0; int64_t b_lo = 0, b_hi = 0; };
        std::vector<SegPart> parts((size_t)PT);
        parallel_chunks(nseg, PT, [&](int th, int64_t sb, int64_t se) {
            SegPart &sp = parts[(size_t)th];
            sp.ivs.reserve((size_t)(se - sb));
            for (int64_t s = sb; s < se; ++s) {
                const int64_t len = end[s] - start[s];
                if (len < 0) { sp.bad = s; sp.bad_kind = 1; return; }
                if (out_step[s] != 1 && out_step[s] != -1 && out_step[s] != 0) { sp.bad = s; sp.bad_kind = 2; return; }
                if (out_step[s] == 0) sp.has_sums = true;
                // output bounds
                if (len > 0) {
                    const int64_t first = out_off[s], last = out_off[s] + (int64_t)out_step[s] * (len - 1);
                    const int64_t lo = std::min(first, last), hi = std::max(first, last) + (int64_t)(rows - 1) * row_stride[s];
                    if (lo < 0 || hi >= out_elems || row_stride[s] < 0) { sp.bad = s; sp.bad_kind = 3; sp.b_lo = lo; sp.b_hi = hi; return; }
                }
                GatherSeg &g = p->gsegs[(size_t)s];
                g.out_off = out_off[s]; g.row_stride = row_stride[s]; g.len = len; g.step = out_step[s]; g.pad = 0;
                g.hist_off = -1; g.clip_lo = 0; g.clip_hi = 0; g.start = start[s];
                sp.covered += (out_step[s] == 0 ? (len > 0 ? 1 : 0) : len) * rows;
                if (tid[s] < 0 || tid[s] >= ntid || len == 0) continue; // unknown chromosome: zeros (genome_array.py:795-798)
                const int64_t cs = std::max<int64_t>(start[s], 0), ce = std::min<int64_t>(end[s], kMaxPos);
                if (ce <= cs) continue;
                g.clip_lo = cs - start[s];
                g.clip_hi = ce - start[s];
                const int m = mode_of(strand[s]);
                sp.modes |= 1u << m;
                sp.ivs.push_back({tid[s], m, cs, ce});
            }
        });
        for (const SegPart &sp : parts)   // the defect of the lowest segment index, as a serial pass reports it
            if (sp.bad >= 0) {
                const long long b = (long long)sp.bad;
                delete p;
                if (sp.bad_kind == 1) return fail(PC_ERR_ARG, "segment %lld: end < start", b);
                if (sp.bad_kind == 2) return fail(PC_ERR_ARG, "segment %lld: out_step must be +1, -1 or 0 (sum)", b);
                return fail(PC_ERR_ARG, "segment %lld: output slice [%lld,%lld] outside buffer of %lld elements", b, (long long)sp.b_lo, (long long)sp.b_hi, (long long)out_elems);
            }
        for (SegPart &sp : parts) {
            modes |= sp.modes;
            p->covered += sp.covered;
            p->has_sums |= sp.has_sums;
            ivs.insert(ivs.end(), sp.ivs.begin(), sp.ivs.end());
        }
    }
    p->modes = modes;
    int nmodes = 0;
    for (int m = 0; m < kModes; ++m) nmodes += (modes >> m) & 1;
    if (nmodes == 0) nmodes = 1;
    {   // window size (the per-call hard limit of LDS is checked in pc_count)
        unsigned long long iv_len = 0;
        for (const Iv &iv : ivs) iv_len += (unsigned long long)(iv.e - iv.s);
        int64_t budget = 0;
        const int G = choose_window(rows, nmodes, (unsigned long long)ivs.size(), iv_len, e->knobs.tile_g, &budget);
        if ((rows > 1 ? 2 : 4) * (int64_t)nmodes * rows * G > 150 * 1024) { delete p; return fail(PC_ERR_ARG, "pc_plan_create: too many rows (%d) for the LDS window", rows); }
        p->G = G;
    }
    const int G = p->G;

    pclk.lap("plan: segments");
    // ---- islands: union of the queried intervals per (tid, mode)
    // sorted by (contig, mode, start, end): contigs first (a counting pass), then every contig's stretch on its own
    // thread -- the contigs of an annotation are independent
    auto by_contig = [ntid, PT](auto &v, auto tid_of, auto less) {
        typedef typename std::remove_reference<decltype(v)>::type Vec;
        if (PT <= 1 || v.size() < (size_t)(1 << 16)) { std::sort(v.begin(), v.end(), less); return; }
        std::vector<size_t> at((size_t)ntid + 1, 0);
        for (const auto &x : v) at[(size_t)tid_of(x) + 1] += 1;
        for (int t = 0; t < ntid; ++t) at[(size_t)t + 1] += at[(size_t)t];
        Vec tmp(v.size());
        { std::vector<size_t> cur(at.begin(), at.end() - 1); for (const auto &x : v) tmp[cur[(size_t)tid_of(x)]++] = x; }
        v.swap(tmp);
        // heaviest contigs first, dealt round-robin
        std::vector<int> order((size_t)ntid);
        for (int t = 0; t < ntid; ++t) order[(size_t)t] = t;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return at[(size_t)a + 1] - at[(size_t)a] > at[(size_t)b + 1] - at[(size_t)b]; });
        parallel_chunks((int64_t)PT, PT, [&](int th, int64_t, int64_t) {
            for (size_t k = (size_t)th; k < order.size(); k += (size_t)PT) {
                const int t = order[k];
                std::sort(v.begin() + (std::ptrdiff_t)at[(size_t)t], v.begin() + (std::ptrdiff_t)at[(size_t)t + 1], less);
            }
        });
    };
    by_contig(ivs, [](const Iv &a) { return a.tid; }, [](const Iv &a, const Iv &b) {
        if (a.tid != b.tid) return a.tid < b.tid;
        if (a.mode != b.mode) return a.mode < b.mode;
        if (a.s != b.s) return a.s < b.s;
        return a.e < b.e;
    });
    struct Island { int32_t tid, mode; int64_t s, e, off; };
    std::vector<Island> islands;
    for (const Iv &iv : ivs) {
        if (!islands.empty() && islands.back().tid == iv.tid && islands.back().mode == iv.mode && iv.s <= islands.back().e)
            islands.back().e = std::max(islands.back().e, iv.e);
        else
            islands.push_back({iv.tid, iv.mode, iv.s, iv.e, 0});
    }
    int64_t npos = 0;
    for (Island &is : islands) { is.off = npos; npos += is.e - is.s; }
    p->npos = npos;

    pclk.lap("plan: islands");
    // ---- every segment -> its island (binary search)
    parallel_chunks(nseg, PT, [&](int, int64_t sb, int64_t se) {
    size_t hint = 0;   // (the exons of a chain follow each other in the caller's arrays: the search starts where the last one ended)
    for (int64_t s = sb; s < se; ++s) {
        GatherSeg &g = p->gsegs[(size_t)s];
        if (g.clip_hi <= g.clip_lo) continue;
        const int64_t cs = start[s] + g.clip_lo;
        const int m = mode_of(strand[s]);
        const int32_t ts = tid[s];
        // first island after the last one with (tid, mode, s) <= (tid, m, cs)
        const size_t lo = gallop_lower_bound(islands.size(), hint, [&](size_t k) {
            const Island &is = islands[k];
            return is.tid < ts || (is.tid == ts && (is.mode < m || (is.mode == m && is.s <= cs)));
        });
        hint = lo;
        const Island &is = islands[lo - 1];
        g.hist_off = is.off + (cs - is.s);
    }
    });

    pclk.lap("plan: segment->island");
    // ---- pieces: islands cut at the fixed genome grid of G positions; tiles: grid windows
    struct RawPiece { int32_t tid; int64_t win; Piece pc_; };
    PodVec<RawPiece> raw;
    // The tables only the center rule and the coordinate export read -- the 64-position chunks, the per-segment gather
    // list -- are built (and uploaded) when first needed for a large annotation: a point-rule plan of 479 k exons
    // otherwise pays for 1.4 M chunk descriptors it never uses.  Small plans keep everything in their one upload.
    p->lazy_center = nseg >= (1 << 16);
    auto cut_island = [G](const Island &is, RawPiece *dst) {   // the island's pieces, in order; returns their number
        size_t k = 0;
        for (int64_t a = is.s; a < is.e;) {
            const int64_t win = (a / G) * G;
            const int64_t b = std::min<int64_t>(is.e, win + G);
            if (dst) {
                Piece pc_;
                pc_.hist_off = is.off + (a - is.s); pc_.start = (int32_t)a; pc_.len = (int32_t)(b - a); pc_.mode = is.mode; pc_.pad = 0;
                dst[k] = {is.tid, win, pc_};
            }
            ++k;
            a = b;
        }
        return k;
    };
    {   // islands -> pieces: counted, placed by prefix sum, filled by all threads
        std::vector<size_t> at(islands.size() + 1, 0);
        parallel_chunks((int64_t)islands.size(), PT, [&](int, int64_t ib, int64_t ie) {
            for (int64_t i = ib; i < ie; ++i) {
                const Island &is = islands[(size_t)i];
                at[(size_t)i + 1] = is.e > is.s ? (size_t)((is.e - 1) / G - is.s / G + 1) : 0;
            }
        });
        for (size_t i = 0; i < islands.size(); ++i) at[i + 1] += at[i];
        raw.resize(at.back());
        parallel_chunks((int64_t)islands.size(), PT, [&](int, int64_t ib, int64_t ie) {
            for (int64_t i = ib; i < ie; ++i) cut_island(islands[(size_t)i], raw.data() + at[(size_t)i]);
        });
        // sorted by (contig, window, mode, start).  The islands are in contig order, so the pieces of a contig are
        // already one stretch of `raw`: every stretch is sorted in place on its own thread, heaviest contigs first
        auto less = [](const RawPiece &a, const RawPiece &b) {
            if (a.win != b.win) return a.win < b.win;
            if (a.pc_.mode != b.pc_.mode) return a.pc_.mode < b.pc_.mode;
            return a.pc_.start < b.pc_.start;
        };
        std::vector<std::pair<size_t, size_t>> stretch;   // [begin, end) of every contig that has pieces
        for (size_t i = 0; i < islands.size();) {
            size_t j = i + 1;
            while (j < islands.size() && islands[j].tid == islands[i].tid) ++j;
            if (at[j] > at[i]) stretch.push_back({at[i], at[j]});
            i = j;
        }
        std::sort(stretch.begin(), stretch.end(), [](const std::pair<size_t, size_t> &a, const std::pair<size_t, size_t> &b) {
            return a.second - a.first > b.second - b.first;
        });
        parallel_chunks((int64_t)PT, PT, [&](int th, int64_t, int64_t) {
            for (size_t k = (size_t)th; k < stretch.size(); k += (size_t)PT)
                std::sort(raw.begin() + (std::ptrdiff_t)stretch[k].first, raw.begin() + (std::ptrdiff_t)stretch[k].second, less);
        });
    }
    int max_slots = 1;
    // A multi-row plan (stratified rule: rows x G bins per strand mode) gives every strand mode of a window a tile of its
    // own: the LDS window of the launch is then sized for ONE mode -- C5: 14 KB instead of 25, seven workgroups per CU
    // instead of six -- and the two windows in a hundred that query both strands are scanned twice.  Single-row plans keep
    // all modes of a window in one tile: one pass over the records serves both strands.
    const bool split_modes = rows > 1;
    auto new_tile = [&](size_t i) {
        return i == 0 || raw[i - 1].tid != raw[i].tid || raw[i - 1].win != raw[i].win || (split_modes && raw[i - 1].pc_.mode != raw[i].pc_.mode);
    };
    auto fill_tile = [&](Tile &t, size_t i0, size_t i1) {   // the tile of the sorted pieces [i0, i1)
        t.tid = raw[i0].tid; t.win_start = (int32_t)raw[i0].win; t.piece_begin = (uint32_t)i0; t.piece_end = (uint32_t)i1;
        t.mode_mask = 0; t.op_begin = t.op_end = 0; t.span_lo = 0xffff; t.span_hi = 0;
        for (size_t i = i0; i < i1; ++i) {
            t.mode_mask |= 1u << raw[i].pc_.mode;
            t.span_lo = std::min<uint16_t>(t.span_lo, (uint16_t)(raw[i].pc_.start - t.win_start));
            t.span_hi = std::max<uint16_t>(t.span_hi, (uint16_t)(raw[i].pc_.start - t.win_start + raw[i].pc_.len));
        }
    };
    if (p->lazy_center && PT > 1) {
        // large plans: every thread takes a stretch of the sorted pieces and owns the tiles that START in it
        p->pieces.resize(raw.size());
        std::vector<size_t> tcount((size_t)PT + 1, 0);
        parallel_chunks((int64_t)raw.size(), PT, [&](int th, int64_t ib, int64_t ie) {
            size_t n = 0;
            for (int64_t i = ib; i < ie; ++i) {
                p->pieces[(size_t)i] = raw[(size_t)i].pc_;
                n += new_tile((size_t)i) ? 1 : 0;
            }
            tcount[(size_t)th + 1] = n;
        });
        for (int th = 0; th < PT; ++th) tcount[(size_t)th + 1] += tcount[(size_t)th];
        p->tiles.resize(tcount[(size_t)PT]);
        parallel_chunks((int64_t)raw.size(), PT, [&](int th, int64_t ib, int64_t ie) {
            size_t k = tcount[(size_t)th];
            for (int64_t i = ib; i < ie; ++i) {
                if (!new_tile((size_t)i)) continue;
                size_t j = (size_t)i + 1;
                while (j < raw.size() && !new_tile(j)) ++j;   // (a tile may end in the next thread's stretch)
                fill_tile(p->tiles[k++], (size_t)i, j);
            }
        });
    } else {
        p->pieces.reserve(raw.size());
        if (!p->lazy_center) p->cchunks.reserve((size_t)(npos / kWave) + raw.size());
        for (size_t i = 0; i < raw.size();) {
            size_t j = i + 1;
            while (j < raw.size() && !new_tile(j)) ++j;
            Tile t;
            fill_tile(t, i, j);
            p->tiles.push_back(t);
            for (; i < j; ++i) {
                p->pieces.push_back(raw[i].pc_);
                // 64-position chunks for the ordered center replay (one wave each)
                for (int32_t a = 0; !p->lazy_center && a < raw[i].pc_.len; a += kWave) {
                    CenterChunk c;
                    c.hist_off = raw[i].pc_.hist_off + a; c.tid = raw[i].tid; c.start = raw[i].pc_.start + a;
                    c.len = std::min<int32_t>(kWave, raw[i].pc_.len - a); c.mode = raw[i].pc_.mode;
                    c.op_begin = (uint32_t)(p->tiles.size() - 1); c.op_end = 0;   // the tile, until its output pieces are known (below)
                    p->cchunks.push_back(c);
                }
            }
        }
    }
    for (const Tile &t : p->tiles) max_slots = std::max(max_slots, __builtin_popcount(t.mode_mask));
    p->max_slots = max_slots;

    pclk.lap("plan: pieces+tiles+chunks");
    // ---- output pieces: every queried segment cut at the tile grid, in the caller's layout
    {
        struct RawOut { uint32_t tile; OutPiece o; };
        std::vector<std::vector<RawOut>> part((size_t)PT);
        std::vector<uint8_t> part_zero((size_t)PT, 0);
        parallel_chunks(nseg, PT, [&](int th, int64_t sb, int64_t se) {
            std::vector<RawOut> &mine = part[(size_t)th];
            mine.reserve((size_t)(se - sb) + (size_t)(se - sb) / 2);
            size_t seg_hint = 0;   // tile of the previous segment's last window: the next exon of the chain is close by
            for (int64_t s = sb; s < se; ++s) {
                const GatherSeg &g = p->gsegs[(size_t)s];
                if (g.len > 0 && (g.hist_off < 0 || g.clip_lo > 0 || g.clip_hi < g.len)) part_zero[(size_t)th] = 1;
                if (g.hist_off < 0 || g.clip_hi <= g.clip_lo) continue;
                const int m = mode_of(strand[s]);
                const int64_t cs = start[s] + g.clip_lo, ce = start[s] + g.clip_hi;
                size_t prev = (size_t)-1;   // tile of the segment's previous window: the next window's tile follows it
                for (int64_t a = cs; a < ce;) {
                    const int64_t win = (a / G) * G;
                    const int64_t b = std::min<int64_t>(ce, win + G);
                    // tile of (tid, win)
                    // (tiles are sorted by contig, window and -- when every mode has its own tile -- mode: a one-mode
                    // tile's mask, 1 << mode, orders like the mode)
                    size_t lo = prev + 1;
                    const uint32_t want_mask = 1u << m;
                    if (prev == (size_t)-1 || lo >= p->tiles.size() || p->tiles[lo].tid != tid[s] || (int64_t)p->tiles[lo].win_start != win ||
                        (split_modes && p->tiles[lo].mode_mask != want_mask)) {
                        const int32_t ts = tid[s];
                        lo = gallop_lower_bound(p->tiles.size(), seg_hint, [&](size_t k) {
                            const Tile &t = p->tiles[k];
                            if (t.tid != ts) return t.tid < ts;
                            if ((int64_t)t.win_start != win) return (int64_t)t.win_start < win;
                            return split_modes && t.mode_mask < want_mask;
                        });
                    }
                    prev = lo;
                    seg_hint = lo;
                    OutPiece o;
                    o.out_off = g.out_off + (int64_t)g.step * (a - start[s]);
                    o.row_stride = g.row_stride;
                    o.hist_off = g.hist_off + (a - cs);
                    o.start = (int32_t)a; o.len = (int32_t)(b - a); o.mode = m; o.step = g.step;
                    mine.push_back({(uint32_t)lo, o});
                    a = b;
                }
            }
        });
        for (uint8_t z : part_zero) if (z) p->out_needs_zero = true;
        // stable counting sort by tile (the thread lists are in segment order, taken in thread order).  The tile
        // index space is cut into one stretch per thread; every thread walks all the lists' tile indices (4 bytes per
        // record) and counts, then places, the records of its stretch: no serial pass over the records.
        const size_t ntl = p->tiles.size();
        std::vector<size_t> list_off(part.size() + 1, 0);
        for (size_t k = 0; k < part.size(); ++k) list_off[k + 1] = list_off[k] + part[k].size();
        PodVec<uint32_t> tile_of(list_off.back());
        parallel_chunks((int64_t)part.size(), PT, [&](int, int64_t kb, int64_t ke) {
            for (int64_t k = kb; k < ke; ++k)
                for (size_t i = 0; i < part[(size_t)k].size(); ++i) tile_of[list_off[(size_t)k] + i] = part[(size_t)k][i].tile;
        });
        std::vector<uint32_t> at(ntl + 1, 0);
        std::vector<size_t> stretch_total((size_t)PT + 1, 0);
        auto stretch = [&](int th, size_t &t0, size_t &t1) { t0 = ntl * (size_t)th / (size_t)PT; t1 = ntl * ((size_t)th + 1) / (size_t)PT; };
        parallel_chunks((int64_t)PT, PT, [&](int th, int64_t, int64_t) {   // records per tile of the stretch
            size_t t0, t1;
            stretch(th, t0, t1);
            size_t n = 0;
            for (const uint32_t t : tile_of)
                if (t >= t0 && t < t1) { at[(size_t)t + 1] += 1; ++n; }
            stretch_total[(size_t)th + 1] = n;
        });
        for (int th = 0; th < PT; ++th) stretch_total[(size_t)th + 1] += stretch_total[(size_t)th];
        p->opieces.resize(list_off.back());
        parallel_chunks((int64_t)PT, PT, [&](int th, int64_t, int64_t) {   // offsets of the stretch's tiles, then placement
            size_t t0, t1;
            stretch(th, t0, t1);
            if (t0 == t1) return;
            uint32_t run = (uint32_t)stretch_total[(size_t)th];
            for (size_t t = t0; t < t1; ++t) {
                const uint32_t c = at[t + 1];
                p->tiles[t].op_begin = run;
                at[t + 1] = run;            // cursor of tile t (at[] is indexed t + 1 within the stretch; at[t0] belongs to the stretch below)
                run += c;
                p->tiles[t].op_end = run;
            }
            for (size_t k = 0; k < part.size(); ++k) {
                const uint32_t *tl = tile_of.data() + list_off[k];
                const std::vector<RawOut> &v = part[k];
                for (size_t i = 0; i < v.size(); ++i)
                    if (tl[i] >= t0 && tl[i] < t1) p->opieces[at[(size_t)tl[i] + 1]++] = v[i].o;
            }
        });
    }

    for (CenterChunk &c : p->cchunks) {   // (eager tables of a small plan) the chunk's output pieces = those of its tile
        const Tile &t = p->tiles[c.op_begin];
        c.op_begin = t.op_begin;
        c.op_end = t.op_end;
    }
    pclk.lap("plan: output pieces");
    // ---- gather work list (center rule)
    for (int64_t s = 0; !p->lazy_center && s < nseg; ++s) {
        const int64_t len = p->gsegs[(size_t)s].len;
        for (int64_t c = 0; c * kGatherChunk < len; ++c) p->gchunks.push_back({(uint32_t)s, (uint32_t)c});
    }

    pclk.lap("plan: gather list");
    // ---- one device block and one upload for all tables (a plan of one short segment is otherwise
    // dominated by the per-copy cost); the per-tile item counters arrive zeroed with it
    size_t bytes = 0;
    auto place = [&bytes](size_t n) { const size_t at = bytes; bytes += (n + 255) & ~(size_t)255; return at; };
    const size_t at_tiles = place(p->tiles.size() * sizeof(Tile)), at_pieces = place(p->pieces.size() * sizeof(Piece)),
                 at_opieces = place(p->opieces.size() * sizeof(OutPiece)), at_cchunks = place(p->cchunks.size() * sizeof(CenterChunk)),
                 at_gsegs = place(p->lazy_center ? 0 : p->gsegs.size() * sizeof(GatherSeg)), at_gchunks = place(p->gchunks.size() * sizeof(GatherChunk)),
                 at_items = place((p->tiles.size() + 1) * sizeof(uint32_t)), at_wcounters = place(64), at_total = place(64);
    // a short compact histogram rides along, already zeroed (one memset less on the first count)
    const size_t hist_full = (size_t)p->npos * (size_t)p->rows * sizeof(double);
    const bool hist_here = hist_full > 0 && hist_full <= 64 * 1024;
    const size_t at_hist = hist_here ? place(hist_full) : 0;
    // (a large annotation's tables -- tens of MB -- go up table by table from where they are: a
    // page-locked buffer of that size costs more to create than it saves)
    const bool through_pinned = bytes <= ((size_t)4 << 20);
    int rc = p->d_tables.reserve(bytes);
    if (rc == PC_OK && through_pinned && e->pinned_busy && hipEventSynchronize(e->ev_pinned) != hipSuccess) rc = fail(PC_ERR_HIP, "pc_plan_create: wait failed");
    if (rc == PC_OK && through_pinned) rc = e->pinned.reserve(bytes);
    if (rc == PC_OK) {
        uint8_t *h = through_pinned ? e->pinned.p : nullptr, *d = p->d_tables.p;
        bool copy_failed = false;
        auto put = [&](size_t at, const void *src, size_t n) {
            if (!n) return;
            if (through_pinned) memcpy(h + at, src, n);
            else if (hipMemcpyAsync(d + at, src, n, hipMemcpyHostToDevice, e->stream) != hipSuccess) copy_failed = true;
        };
        put(at_tiles, p->tiles.data(), p->tiles.size() * sizeof(Tile));
        put(at_pieces, p->pieces.data(), p->pieces.size() * sizeof(Piece));
        put(at_opieces, p->opieces.data(), p->opieces.size() * sizeof(OutPiece));
        put(at_cchunks, p->cchunks.data(), p->cchunks.size() * sizeof(CenterChunk));
        if (!p->lazy_center) put(at_gsegs, p->gsegs.data(), p->gsegs.size() * sizeof(GatherSeg));
        put(at_gchunks, p->gchunks.data(), p->gchunks.size() * sizeof(GatherChunk));
        if (through_pinned) memset(h + at_items, 0, bytes - at_items);
        else if (hipMemsetAsync(d + at_items, 0, bytes - at_items, e->stream) != hipSuccess) copy_failed = true;
        p->d_tiles.p = (Tile *)(d + at_tiles); p->d_pieces.p = (Piece *)(d + at_pieces);
        p->d_opieces.p = (OutPiece *)(d + at_opieces); p->d_cchunks.p = (CenterChunk *)(d + at_cchunks);
        p->d_gsegs.p = (GatherSeg *)(d + at_gsegs); p->d_gchunks.p = (GatherChunk *)(d + at_gchunks);
        p->d_tile_items.p = (uint32_t *)(d + at_items); p->d_total.p = d + at_total;
        p->d_wcounters.p = (uint32_t *)(d + at_wcounters);
        p->tile_items_zero = true;
        p->wcounters_zero = true;
        if (hist_here) { p->d_hist.p = d + at_hist; p->hist_kind = 0; p->hist_clean = true; }
        if (!through_pinned) {   // the copies read the plan's own vectors, which live as long as the plan
            if (copy_failed) rc = fail(PC_ERR_HIP, "pc_plan_create: upload failed");
        } else if (hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, e->stream) != hipSuccess) rc = fail(PC_ERR_HIP, "pc_plan_create: upload failed");
        else if (hipEventRecord(e->ev_pinned, e->stream) != hipSuccess) rc = fail(PC_ERR_HIP, "pc_plan_create: event failed");
        else e->pinned_busy = true;
    }
    pclk.lap("plan: upload");
    p->n_tiles = p->tiles.size(); p->n_pieces = p->pieces.size(); p->n_opieces = p->opieces.size();
    p->n_cchunks = p->cchunks.size(); p->n_gchunks = p->gchunks.size();
    if (rc != PC_OK) {
        (void)hipStreamSynchronize(e->stream);   // copies out of the plan's vectors may be in flight
        delete p;
        return rc;
    }
    *out = p;
    return PC_OK;
}

int pc_plan_destroy(pc_plan *p) {
    if (!p) return PC_OK;
    if (p->e) {
        (void)hipSetDevice(p->e->device);
        (void)hipStreamSynchronize(p->e->stream);
    }
    delete p;
    return PC_OK;
}

int64_t pc_plan_positions(pc_plan *p) { return p ? p->npos : -1; }
int64_t pc_plan_tiles(pc_plan *p) { return p ? (int64_t)p->n_tiles : -1; }

int pc_plan_table(pc_plan *p, int which, void *buf, int64_t cap_bytes, int64_t *bytes) {
    if (!p || !bytes || which < 0 || which > 4 || cap_bytes < 0 || (cap_bytes > 0 && !buf)) return fail(PC_ERR_ARG, "pc_plan_table: bad arguments");
    HIP_TRY(hipSetDevice(p->e->device));
    HIP_TRY(hipStreamSynchronize(p->e->stream));
    if (which == 4) {
        const int64_t v[12] = {p->G, (int64_t)p->modes, p->max_slots, p->npos, p->covered, p->has_sums ? 1 : 0, p->out_needs_zero ? 1 : 0,
                               (int64_t)p->n_tiles, (int64_t)p->n_pieces, (int64_t)p->n_opieces, p->gpu_built ? 1 : 0, p->nseg};
        *bytes = (int64_t)sizeof(v);
        std::memcpy(buf, v, (size_t)std::min<int64_t>(cap_bytes, *bytes));
        return PC_OK;
    }
    const void *src = nullptr;
    size_t n = 0;
    bool on_host = false;
    if (which == 0) { src = p->d_tiles.p; n = p->n_tiles * sizeof(Tile); }
    else if (which == 1) { src = p->d_pieces.p; n = p->n_pieces * sizeof(Piece); }
    else if (which == 2) { src = p->d_opieces.p; n = p->n_opieces * sizeof(OutPiece); }
    else if (p->gpu_built) { src = p->d_gsegs_own.p; n = (size_t)p->nseg * sizeof(GatherSeg); }
    else { src = p->gsegs.data(); n = p->gsegs.size() * sizeof(GatherSeg); on_host = true; }
    *bytes = (int64_t)n;
    const size_t take = (size_t)std::min<int64_t>(cap_bytes, (int64_t)n);
    if (take) {
        if (on_host) std::memcpy(buf, src, take);
        else HIP_TRY(hipMemcpy(buf, src, take, hipMemcpyDeviceToHost));
    }
    return PC_OK;
}

int pc_count(pc_engine *e, pc_plan *p, int out_dtype) {
    if (!e || !p || p->e != e) return fail(PC_ERR_ARG, "pc_count: bad engine/plan");
    if (!e->have_map) return fail(PC_ERR_STATE, "pc_count: no mapping rule set (pc_set_mapping)");
    if (e->files.empty()) return fail(PC_ERR_STATE, "pc_count: no alignments staged (pc_add_alignment_file)");
    if (out_dtype != PC_OUT_INT64 && out_dtype != PC_OUT_FLOAT64) return fail(PC_ERR_ARG, "pc_count: bad out_dtype");
    if (p->rows != e->rows) return fail(PC_ERR_ARG, "pc_count: plan built for %d rows, mapping rule has %d", p->rows, e->rows);
    if (e->ff_max_nh)
        for (size_t f = 0; f < e->files.size(); ++f)
            if (e->files[f]->n > 0 && !e->files[f]->have_nh)
                return fail(PC_ERR_STATE, "pc_count: an NH filter is set but alignment file %d has no NH column (pc_set_alignment_nh)", (int)f);
    if (e->ff_on)   // a file staged after pc_set_flag_filter gets its verdicts when its columns arrive: not counted without them
        for (size_t f = 0; f < e->files.size(); ++f)
            if (e->files[f]->n > 0 && !e->files[f]->have_sam)
                return fail(PC_ERR_STATE, "pc_count: a FLAG / MAPQ filter is set but alignment file %d has no FLAG / MAPQ columns (pc_set_alignment_sam)", (int)f);
    const bool center = e->kind == PC_MAP_CENTER;
    if ((center || e->norm_on) && out_dtype != PC_OUT_FLOAT64)
        return fail(PC_ERR_ARG, "pc_count: center mapping / normalisation produce float64 (map_factories.pyx:230, genome_array.py:826-827)");
    HIP_TRY(hipSetDevice(e->device));
    int rc = PC_OK;
    if (center) {   // the center-only tables of a large plan; the center streams of the strand selections it queries
        rc = ensure_center_tables(e, p);
        if (rc != PC_OK) return rc;
        const bool need[3] = {(p->modes & 1u) != 0, (p->modes & 2u) != 0, (p->modes & 12u) != 0};
        for (auto *f : e->files)
            for (int k = 0; k < 3 && rc == PC_OK; ++k)
                if (need[k]) rc = build_center_stream(e, f, k, e->param);   // (no-op when built for this nibble)
        if (rc != PC_OK) return rc;
    }
    rc = refresh_file_views(e);
    if (rc != PC_OK) return rc;

    // the compact histogram (uint32 per island position and row): what merged windows of the point rules go through;
    // the center rule does not use it
    const size_t hist_bytes = center ? 0 : (size_t)p->npos * p->rows * sizeof(uint32_t);
    if (!center && !p->d_hist.p) {
        rc = p->d_hist_own.reserve(std::max<size_t>((size_t)p->npos * p->rows * sizeof(uint32_t), 8));
        p->d_hist.p = p->d_hist_own.p;
    }
    if (rc == PC_OK) rc = p->d_out.reserve(std::max<size_t>((size_t)p->out_elems * 8, 8));
    if (rc != PC_OK) return rc;

    const int nfiles = (int)e->files.size();
    const int W = e->W();
    const int G = p->G;
    const int64_t R = e->knobs.work_r;                             // records per work item
    const int64_t pile = e->knobs.pile ? e->knobs.pile : 12 * R;   // a 128-nt sub-window with more records than this is merged through the histogram
    const MapParams mp = e->params();
    const int ntiles = (int)p->n_tiles;
    hipStream_t st = e->stream;
    int64_t nrec = 0, nextra = 0;
    for (auto *f : e->files) { nrec += f->n; nextra += f->nrun; }

    if (e->prof_level >= 1) HIP_TRY(hipEventRecord(e->ev[0], st));
    // Outputs are written exactly once by the tile kernels.  Only positions that belong to no
    // tile (unknown contig, clipped coordinates) or gaps the caller left between slices need a
    // zero fill; the compact histogram of the point rules is kept all-zero between calls.
    if (p->has_sums && (center || e->norm_on))
        return fail(PC_ERR_ARG, "pc_count: summed slices (out_step 0) need an integer mapping rule without normalisation");
    if ((p->has_sums || p->out_needs_zero || p->covered != p->out_elems) && p->out_elems)
        HIP_TRY(hipMemsetAsync(p->d_out.p, 0, (size_t)p->out_elems * 8, st));
    // (a large histogram is not cleared as a whole: k_clear_split zeroes the slices of the merged windows behind every
    // k_tile_ranges, which is all that is ever read of it)
    p->hist_lazy = !center && (int64_t)hist_bytes >= e->knobs.hist_lazy_bytes && !e->knobs.hist_memset;
    if (!center && hist_bytes && !(p->hist_clean && p->hist_kind == 0) && !p->hist_lazy) {
        HIP_TRY(hipMemsetAsync(p->d_hist.p, 0, hist_bytes, st));
    }
    if (e->prof_level >= 2) HIP_TRY(hipEventRecord(e->ev[1], st));

    if (!center) {
        p->hist_kind = 0;
        p->hist_clean = true; // k_gather_split clears what the split tiles merged
        // ---- a plan of ONE window over one file (`ga[segment]`): the whole count is one launch -- the workgroup looks its
        // record ranges up itself; no work list, no second window class, no merge pass, no events
        // (not under the stratified rule: its 16-bit bins rely on the work lists, which cut or merge a window that scans
        // more than 65 535 records)
        const bool single = ntiles == 1 && nfiles == 1 && !e->knobs.debug_work && !e->knobs.no_single && e->kind != PC_MAP_STRAT5;
        if (single) {
            int lmin = 65536, lmax = -1;
            for (auto *f : e->files) { lmin = std::min(lmin, f->len_min); lmax = std::max(lmax, f->len_max); }
            int tab_lo = 0, tab_n = 0;
            if ((e->kind == PC_MAP_VAR5 || e->kind == PC_MAP_STRAT5) && lmax >= lmin) {
                tab_lo = lmin;
                tab_n = std::max(0, std::min(std::min(lmax, e->table_len - 1) - lmin + 1, 1024));
            }
            int fast_lo = kStreamMaxLen, fast_hi = 0;
            for (auto *f : e->files) { fast_lo = std::min(fast_lo, f->tlen_min); fast_hi = std::max(fast_hi, f->tlen_max); }
            fast_lo = std::min(fast_lo, fast_hi);
            const size_t stage_words = (size_t)kOpStage * sizeof(OutPiece) / sizeof(uint32_t);
            const size_t lds = ((size_t)p->max_slots * p->rows * G + (size_t)((tab_n + 3) & ~3) + stage_words + (size_t)(fast_hi + 1) * kModes + 64) * sizeof(uint32_t);
            if (lds > e->max_lds)
                return fail(PC_ERR_ARG, "pc_count: the window needs %zu bytes of LDS, the device offers %zu per workgroup (too many rows)", lds, e->max_lds);
            const FileView fv0 = e->files[0]->view();
            const int outmode = e->norm_on ? 2 : (out_dtype == PC_OUT_FLOAT64 ? 1 : 0);
            if (e->prof_level >= 1) HIP_TRY(hipEventRecord(e->ev[2], st));
#define PC_LAUNCH_SINGLE(K, O)                                                                                        \
    hipLaunchKernelGGL((k_hist_point<K, O, kHistWG, false, false, true>), dim3(1), dim3(kHistWG), lds, st, p->d_pieces.p, p->d_opieces.p, \
                       fv0, fv0, e->d_files.p, (const WorkItem *)p->d_tiles.p, p->d_wcounters.p, p->d_tile_items.p, mp, G, p->max_slots,    \
                       tab_lo, tab_n, fast_lo, fast_hi, (uint32_t *)p->d_hist.p, (int64_t)e->Ws(), (OutT_<O>::type *)p->d_out.p,           \
                       e->norm_sum, (uint32_t)e->Wg(), (uint32_t)e->Wr(), (const FileRange *)nullptr, nfiles, Tile{}, OutPiece{}, (uint32_t *)nullptr, 0u)
#define PC_LAUNCH_SINGLE_O(K)                                                                                         \
    do {                                                                                                              \
        if (outmode == 0) PC_LAUNCH_SINGLE(K, 0);                                                                     \
        else if (outmode == 1) PC_LAUNCH_SINGLE(K, 1);                                                                \
        else PC_LAUNCH_SINGLE(K, 2);                                                                                  \
    } while (0)
            switch (e->kind) {
            case PC_MAP_FIVE: PC_LAUNCH_SINGLE_O(0); break;
            case PC_MAP_THREE: PC_LAUNCH_SINGLE_O(1); break;
            case PC_MAP_VAR5: PC_LAUNCH_SINGLE_O(3); break;
            default: PC_LAUNCH_SINGLE_O(4); break;
            }
#undef PC_LAUNCH_SINGLE_O
#undef PC_LAUNCH_SINGLE
            if (e->prof_level >= 1) HIP_TRY(hipEventRecord(e->ev[3], st));
            if (e->prof_level >= 2) HIP_TRY(hipEventRecord(e->ev[4], st));
        } else if (ntiles > 0) {
            // work list capacity: every record lies in at most 1 + ceil(W/G) scan windows
            // work-list capacity (an upper bound): a window scanning n records yields at most
            // max(1, 2n/R) items, and every record is scanned by at most 1 + (W+127)/G windows
            const int halo = std::max(std::max(W, e->Ws()), std::max(e->Wg(), e->Wr()));
            // (a multi-row plan gives every strand mode of a window a tile of its own -- pc_plan_create, split_modes --
            // so a record is scanned by up to popcount(modes) tiles per window)
            const int tiles_per_window = p->rows > 1 ? std::max(1, __builtin_popcount(p->modes)) : 1;
            const double windows_per_record = (1.0 + (double)(halo + 127) / (double)G) * (double)tiles_per_window;
            int64_t cap64 = (int64_t)ntiles * nfiles + (int64_t)(2.0 * windows_per_record * (double)nrec / (double)R) + nfiles + 64;
            if (e->kind == PC_MAP_STRAT5) {
                // 16-bit bins: a window that scans more than 65 535 records, runs and list entries is merged, in slices of ONE
                // kind of range each (k_tile_ranges): at most adds / R + 4 items per such window, and fewer than adds / 65 535 of them
                int64_t nxl = 0, ngp = 0;
                for (auto *f : e->files) { nxl += f->nxlong; ngp += f->ngap; }
                cap64 += (int64_t)(6.0 * windows_per_record * (double)(nrec + nextra + ngp) / (double)R) +
                         (int64_t)ntiles * nfiles * (4 + 2 * (nxl / R));
            }
            if (cap64 >= (int64_t)0xffffffffu) return fail(PC_ERR_ARG, "pc_count: work list too large");
            rc = p->d_work.reserve((size_t)cap64);
            if (rc != PC_OK) return rc;
            // sparse windows: single-wave workgroups with a small LDS window (rows == 1 only)
            // (skipped for dense annotations, where queried positions fill most of every window)
            const bool sparse_plan = (double)p->npos < 0.25 * (double)ntiles * (double)G;
            const int small_g = ((p->rows == 1 || e->knobs.small_rows) && sparse_plan && !e->knobs.no_small) ? std::min(e->knobs.small_g, G) : 0;
            const int64_t small_n = e->knobs.small_n;
            const int64_t cap_small = small_g ? (int64_t)ntiles * nfiles : 0;
            rc = p->d_work_small.reserve((size_t)std::max<int64_t>(cap_small, 1));
            if (rc == PC_OK && nfiles > 1) rc = p->d_chain.reserve((size_t)cap64 * (size_t)(nfiles - 1));
            if (rc == PC_OK && nfiles > 1) rc = p->d_chain_small.reserve((size_t)std::max<int64_t>(cap_small, 1) * (size_t)(nfiles - 1));
            if (rc != PC_OK) return rc;
            pc_plan::WorkKey key;
            key.generation = e->work_generation; key.nfiles = nfiles; key.G = G; key.Wg = e->Wg(); key.Ws = e->Ws(); key.Wr = e->Wr();
            key.small_g = small_g; key.R = R; key.pile = pile; key.small_n = small_n; key.cap = cap64;
            if (!(p->work_valid && p->work_key == key) || e->knobs.debug_work) {
                // the lists of this plan are (re)built: counters and per-tile item counts start from zero (they arrive
                // zeroed with the plan's tables, so the first count of a plan needs no memset)
                if (!p->wcounters_zero) HIP_TRY(hipMemsetAsync(p->d_wcounters.p, 0, 8 * sizeof(uint32_t), st));
                if (!p->tile_items_zero) HIP_TRY(hipMemsetAsync(p->d_tile_items.p, 0, ((size_t)ntiles + 1) * sizeof(uint32_t), st));
                p->wcounters_zero = false;
                p->tile_items_zero = false;
                const int64_t nwin = nfiles > 1 ? (int64_t)ntiles : (int64_t)ntiles * nfiles; // one thread per window (several files: joint windows)
                // ... or sixteen lanes per window while that still fits the chip at once: the exact record bounds of a window are
                // then searched by the group (three rounds of sixteen probes instead of a dozen dependent loads each) -- a plan of
                // a few thousand windows (C2: 6 144) otherwise runs on two dozen CUs at the pace of one thread's load chain
                const bool group16 = nwin * 16 <= e->knobs.ranges_cg16_max && !e->knobs.ranges_cg1;
                const int64_t nthreads = group16 ? nwin * 16 : nwin;
#define PC_LAUNCH_RANGES(CG)                                                                                           \
    hipLaunchKernelGGL((k_tile_ranges<CG>), dim3((unsigned)((nthreads + kRangesWG - 1) / kRangesWG)), dim3(kRangesWG), 0, st, p->d_tiles.p, ntiles, \
                       e->files[0]->view(), e->d_files.p, nfiles, G, e->Wg(), e->Ws(), e->Wr(), R, pile, p->d_work.p, p->d_wcounters.p, p->d_tile_items.p, (uint32_t)cap64, \
                       p->d_work_small.p, small_g, small_n, e->knobs.debug_work, p->d_chain.p, p->d_chain_small.p, e->kind == PC_MAP_STRAT5 ? 1 : 0)
                if (group16) PC_LAUNCH_RANGES(16); else PC_LAUNCH_RANGES(1);
#undef PC_LAUNCH_RANGES
                if (p->hist_lazy)
                    hipLaunchKernelGGL(k_clear_split, dim3((unsigned)((ntiles + kClearPerWG - 1) / kClearPerWG)), dim3(kWG), 0, st, p->d_tiles.p, ntiles,
                                       p->d_pieces.p, p->d_tile_items.p, p->d_wcounters.p, p->rows, (uint32_t *)p->d_hist.p, (int64_t)p->npos);
                p->work_key = key;
                p->work_valid = true;
                p->work_counts_known = false;      // the counts of the lists just replaced size no grid
                p->guard_pending = true;           // k_gather_split checks the new lists against the capacity: read with the results
                p->work_counts_generation = 0;
                // The first count of a plan does not know how many items its lists hold, and used to launch the whole
                // capacity: for a sparse annotation under a multi-row rule that is 3.6 M workgroups for 0.53 M items (C5:
                // 3.88 ms against 3.24, and the merge pass on top).  Where the capacity is far above the window count, the
                // queued counts are read back NOW -- one small copy and a stream synchronisation, ~20 us of idle GPU -- and
                // this count already launches exact grids.
                const int64_t spare = cap64 + cap_small - nwin;
                if (ntiles >= 4096 && spare >= e->knobs.first_sync_spare && !e->knobs.test_stale_counts && !e->knobs.debug_work) {
                    if (!p->h_work_counts) {
                        HIP_TRY(hipHostMalloc((void **)&p->h_work_counts, 8 * sizeof(uint32_t), hipHostMallocDefault));
                        HIP_TRY(hipEventCreateWithFlags(&p->ev_work_counts, hipEventDisableTiming));
                    }
                    HIP_TRY(hipMemcpyAsync(p->h_work_counts, p->d_wcounters.p, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                    HIP_TRY(hipStreamSynchronize(st));
                    for (int k = 0; k < 3; ++k) p->work_counts[k] = p->h_work_counts[k];
                    p->work_merged = p->h_work_counts[4];
                    p->work_counts_known = true;
                    p->work_counts_generation = e->work_generation;
                }
            }
            if (e->prof_level >= 1) HIP_TRY(hipEventRecord(e->ev[2], st));
            // offset tables are staged in LDS for the aligned lengths that occur in the data
            int lmin = 65536, lmax = -1;
            for (auto *f : e->files) { lmin = std::min(lmin, f->len_min); lmax = std::max(lmax, f->len_max); }
            int tab_lo = 0, tab_n = 0;
            if ((e->kind == PC_MAP_VAR5 || e->kind == PC_MAP_STRAT5) && lmax >= lmin) {
                tab_lo = lmin;
                tab_n = std::min(lmax, e->table_len - 1) - lmin + 1;
                tab_n = std::max(0, std::min(tab_n, 1024)); // longer reads look the tables up in HBM
            }
            const size_t stage_words = (size_t)kOpStage * sizeof(OutPiece) / sizeof(uint32_t); // output pieces parked in LDS
            const bool b16 = e->kind == PC_MAP_STRAT5;   // 16-bit bins, two positions per word (k_hist_point)
            const size_t bins_words = (((size_t)p->max_slots * p->rows * G) >> (b16 ? 1 : 0)) + (size_t)((tab_n + 3) & ~3) + stage_words;
            // LDS entry table of the record stream: the aligned lengths the stream carries
            int fast_lo = kStreamMaxLen, fast_hi = 0;
            for (auto *f : e->files) { fast_lo = std::min(fast_lo, f->tlen_min); fast_hi = std::max(fast_hi, f->tlen_max); }
            fast_lo = std::min(fast_lo, fast_hi);
            const size_t fwords = (size_t)(fast_hi + 1) * kModes + 64; // entry table + one dump word per lane (after the staged pieces)
            const size_t lds = (bins_words + fwords) * sizeof(uint32_t);
            if (lds > e->max_lds)
                return fail(PC_ERR_ARG, "pc_count: the window needs %zu bytes of LDS, the device offers %zu per workgroup (too many rows)", lds, e->max_lds);
            const FileView fv0 = e->files[0]->view();
            const FileView fv1 = nfiles > 1 ? e->files[1]->view() : fv0;
            // grids: the whole list capacity, or -- once a count of this plan has shown how many items each
            // class queues (same alignments, same knobs) -- exactly those: a sparse annotation leaves most
            // of the capacity empty, and an empty workgroup still costs a dispatch slot
            unsigned grid = (unsigned)cap64, grid_front = (unsigned)cap64, grid_small = (unsigned)cap_small;
            uint32_t launched[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};   // exact grids: what k_gather_split checks the queued counts against
            const bool track_counts = ntiles >= 4096;
            if (track_counts && p->work_counts_generation == e->work_generation && p->h_work_counts && !p->work_counts_known &&
                hipEventQuery(p->ev_work_counts) == hipSuccess) {
                for (int k = 0; k < 3; ++k) p->work_counts[k] = p->h_work_counts[k];
                p->work_merged = p->h_work_counts[4];
                p->work_counts_known = true;   // deterministic for this plan while the generation stands: no further read-backs
            }
            if (track_counts && p->work_counts_known && p->work_counts_generation == e->work_generation) {
                const uint32_t nh = p->work_counts[0], nl = p->work_counts[1] - ((e->knobs.test_stale_counts && p->work_counts[1]) ? 1u : 0u), ns = p->work_counts[2];
                if ((uint64_t)nh + nl <= (uint64_t)cap64 && (int64_t)ns <= cap_small) {
                    grid_front = nh;
                    grid = std::max(1u, nh + nl);
                    grid_small = ns;
                    launched[0] = nh; launched[1] = nl; launched[2] = ns;
                    p->exact_grid_used = true;
                }
            }
            const int outmode = e->norm_on ? 2 : (out_dtype == PC_OUT_FLOAT64 ? 1 : 0);
#define PC_LAUNCH_HIST(K, O)                                                                                          \
    do {                                                                                                              \
        if (nfiles > 1) PC_LAUNCH_HIST_M(K, O, true); else PC_LAUNCH_HIST_M(K, O, false);                             \
    } while (0)
#define PC_LAUNCH_HIST_M(K, O, M)                                                                                     \
    do {                                                                                                              \
        hipLaunchKernelGGL((k_hist_point<K, O, kHistWG, false, M>), dim3(grid), dim3(kHistWG), lds, st, p->d_pieces.p,             \
                           p->d_opieces.p, fv0, fv1, e->d_files.p, p->d_work.p, p->d_wcounters.p, p->d_tile_items.p, mp, \
                           G, p->max_slots, tab_lo, tab_n, fast_lo, fast_hi, (uint32_t *)p->d_hist.p, p->npos,                        \
                           (OutT_<O>::type *)p->d_out.p,                                                                \
                           e->norm_sum, (uint32_t)cap64, grid_front, p->d_chain.p, nfiles, Tile{}, OutPiece{}, (uint32_t *)nullptr, 0u); \
        if (cap_small && grid_small)                                                                                  \
            hipLaunchKernelGGL((k_hist_point<K, O, 64, true, M>), dim3(grid_small), dim3(64), lds_small, st_small, \
                               p->d_pieces.p, p->d_opieces.p, fv0, fv1, e->d_files.p, p->d_work_small.p,                \
                               p->d_wcounters.p, p->d_tile_items.p, mp, small_g, p->max_slots, tab_lo, tab_n, fast_lo, fast_hi, \
                               (uint32_t *)p->d_hist.p,                                                                 \
                               p->npos, (OutT_<O>::type *)p->d_out.p, e->norm_sum, (uint32_t)cap_small, grid_small,      \
                               p->d_chain_small.p, nfiles, Tile{}, OutPiece{}, (uint32_t *)nullptr, 0u);                  \
    } while (0)
#define PC_LAUNCH_HIST_O(K)                                                                                           \
    do {                                                                                                              \
        if (outmode == 0) PC_LAUNCH_HIST(K, 0);                                                                       \
        else if (outmode == 1) PC_LAUNCH_HIST(K, 1);                                                                  \
        else PC_LAUNCH_HIST(K, 2);                                                                                    \
    } while (0)
            const size_t lds_small = ((((size_t)p->max_slots * p->rows * std::max(small_g, 1)) >> (b16 ? 1 : 0)) + (size_t)((tab_n + 3) & ~3) + fwords + stage_words) * sizeof(uint32_t);
            // sparse windows (single-wave workgroups) and dense ones are two independent launches over
            // disjoint windows: they run side by side on two streams, forked after the work lists exist
            // and joined before the last kernel of the call
            hipStream_t st_small = st;
            if (cap_small) {
                st_small = e->side_stream;
                HIP_TRY(hipEventRecord(e->ev_fork, st));
                HIP_TRY(hipStreamWaitEvent(st_small, e->ev_fork, 0));
            }
            switch (e->kind) {
            case PC_MAP_FIVE: PC_LAUNCH_HIST_O(0); break;
            case PC_MAP_THREE: PC_LAUNCH_HIST_O(1); break;
            case PC_MAP_VAR5: PC_LAUNCH_HIST_O(3); break;
            default: PC_LAUNCH_HIST_O(4); break;
            }
#undef PC_LAUNCH_HIST_O
#undef PC_LAUNCH_HIST
#undef PC_LAUNCH_HIST_M
            if (cap_small) {
                HIP_TRY(hipEventRecord(e->ev_join, st_small));
                HIP_TRY(hipStreamWaitEvent(st, e->ev_join, 0));
            }
            if (e->prof_level >= 1) HIP_TRY(hipEventRecord(e->ev[3], st));
            if (e->prof_level >= 2) HIP_TRY(hipEventRecord(e->ev[4], st));
            // tiles that were split into several work items: lay out from the merged histogram
            const int split_per_wg = kWG; // merged windows are the exception (pile-ups), with several files too (joint windows)
#define PC_LAUNCH_SPLIT(O)                                                                                            \
    hipLaunchKernelGGL((k_gather_split<O>), dim3((unsigned)((ntiles + split_per_wg - 1) / split_per_wg)), dim3(kWG), 0, st, \
                       p->d_tiles.p, ntiles, split_per_wg, p->d_pieces.p,                                               \
                       p->d_opieces.p, p->d_tile_items.p, p->d_wcounters.p, p->rows, (uint32_t *)p->d_hist.p, p->npos,   \
                       (OutT_<O>::type *)p->d_out.p, e->norm_sum, launched[0], launched[1], launched[2], (uint32_t)cap64, e->d_counters.p + 12)
            // (skipped once the plan's lists are known to hold no merged window: the lists are the plan's own and do not
            // change from count to count, so neither does that -- and the exact grids it would check were read from them)
            const bool nothing_to_merge = launched[0] != 0xffffffffu && p->work_merged == 0 && !e->knobs.test_stale_counts;
            if (nothing_to_merge) {}
            else if (outmode == 0) PC_LAUNCH_SPLIT(0);
            else if (outmode == 1) PC_LAUNCH_SPLIT(1);
            else PC_LAUNCH_SPLIT(2);
#undef PC_LAUNCH_SPLIT
            if (track_counts) {   // how many items each class queued (k_gather_split keeps a copy): sizes the next launch
                if (!p->h_work_counts) {
                    HIP_TRY(hipHostMalloc((void **)&p->h_work_counts, 8 * sizeof(uint32_t), hipHostMallocDefault));
                    HIP_TRY(hipEventCreateWithFlags(&p->ev_work_counts, hipEventDisableTiming));
                }
                if (p->work_counts_generation != e->work_generation) {   // one read-back per (plan, generation)
                    p->work_counts_known = false;
                    HIP_TRY(hipMemcpyAsync(p->h_work_counts, p->d_wcounters.p, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                    HIP_TRY(hipEventRecord(p->ev_work_counts, st));
                    p->work_counts_generation = e->work_generation;
                }
            }
            if (e->knobs.debug_work) { // diagnostics: how many work items of each class this call queued
                uint32_t c4[4] = {0, 0, 0, 0};
                HIP_TRY(hipMemcpyAsync(c4, p->d_wcounters.p, sizeof(c4), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                fprintf(stderr, "[work] tiles %d: heavy %u light %u small %u, long-span candidates %u (capacity %lld, G %d, R %lld)\n", ntiles,
                        c4[0], c4[1], c4[2], c4[3], (long long)cap64, G, (long long)R);
            }
        } else {
            if (e->prof_level >= 1) HIP_TRY(hipEventRecord(e->ev[2], st));
            if (e->prof_level >= 1) HIP_TRY(hipEventRecord(e->ev[3], st));
            if (e->prof_level >= 2) HIP_TRY(hipEventRecord(e->ev[4], st));
        }
    } else {
        // (k_center writes every queried position of the tiles straight into the output layout; the compact histogram
        // is not touched)
        if (e->prof_level >= 1) HIP_TRY(hipEventRecord(e->ev[2], st));
        const int64_t nchunks = (int64_t)p->n_cchunks;
        if (nchunks > 0) {
            if (kCenterCap * nchunks >= (int64_t)1 << kSubShift) return fail(PC_ERR_ARG, "pc_count: too many positions for the center rule");
            rc = p->d_corder.reserve((size_t)(kCenterCap * nchunks));   // dispatch list: heavy entries front, light back
            if (rc == PC_OK) rc = p->d_ccand.reserve((size_t)nchunks);
            if (rc == PC_OK) rc = p->d_cranges.reserve((size_t)nchunks * (size_t)nfiles);
            if (rc == PC_OK) rc = p->d_crec.reserve((size_t)nchunks * (size_t)nfiles);
            if (rc == PC_OK) rc = p->d_crows.reserve((size_t)nchunks * (size_t)nfiles * (size_t)(2 * kCenterRows));
            if (rc == PC_OK) rc = p->d_ccounts.reserve(8);
            if (rc == PC_OK) rc = e->d_cvalh.reserve(256);
            // one alignment file (every BASELINE config): descriptors per dispatch entry, several entries per wave (k_center2);
            // several files keep round 4's kernel, whose waves walk the files of a chunk one after the other
            const bool slots_on = true;   // (round 6: descriptors for plans over several files too -- one per entry and file)
            if (rc == PC_OK) rc = p->d_cslots.reserve((size_t)(2 * nchunks) * (size_t)nfiles);   // (heavy entries < chunks, light entries <= chunks)
            if (rc != PC_OK) return rc;
            hipLaunchKernelGGL(k_center_vals, dim3(1), dim3(256), 0, st, mp, e->d_invh.p, e->d_cvalh.p);
            if (p->center_generation != e->work_generation || p->center_W != W || p->center_slots != slots_on || p->center_nfiles != nfiles) {
                p->center_nfiles = nfiles;
                HIP_TRY(hipMemsetAsync(p->d_ccounts.p, 0, 8 * sizeof(uint32_t), st));
                unsigned long long *total = (unsigned long long *)(p->d_ccounts.p + 2);
                const unsigned wgs = (unsigned)((nchunks + kRangesWG - 1) / kRangesWG);
                hipLaunchKernelGGL(k_center_weigh, dim3(wgs), dim3(kRangesWG), 0, st, p->d_cchunks.p, nchunks, e->d_files.p, nfiles, W,
                                   p->d_ccand.p, p->d_cranges.p, p->d_crec.p, p->d_crows.p, total);
                // cut thresholds, in multiples of the mean candidate count
                const int ck1 = e->knobs.center_t1, ck2 = e->knobs.center_t2;
                hipLaunchKernelGGL(k_center_order, dim3(wgs), dim3(kRangesWG), 0, st, p->d_ccand.p, nchunks, total, e->knobs.center_floor,
                                   (int64_t)2048, ck1, ck2, p->d_corder.p, p->d_ccounts.p);
                hipLaunchKernelGGL(k_center_slots, dim3((unsigned)((2 * nchunks + kRangesWG - 1) / kRangesWG)), dim3(kRangesWG), 0, st, p->d_cchunks.p, nchunks,
                                       e->d_files.p, nfiles, W, p->d_corder.p, p->d_ccounts.p, p->d_cranges.p, p->d_crec.p, p->d_crows.p, p->d_opieces.p,
                                       p->d_cslots.p, (unsigned long long *)(p->d_ccounts.p + 4));
                p->center_slots = slots_on;
                p->center_generation = e->work_generation;
                p->center_W = W;
                // how many entries the list got: sizes the grid of the later counts of this plan (read back once)
                if (!p->h_center_counts) {
                    HIP_TRY(hipHostMalloc((void **)&p->h_center_counts, 2 * sizeof(uint32_t), hipHostMallocDefault));
                    HIP_TRY(hipEventCreateWithFlags(&p->ev_center_counts, hipEventDisableTiming));
                }
                p->center_counts_known = false;
                HIP_TRY(hipMemcpyAsync(p->h_center_counts, p->d_ccounts.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipEventRecord(p->ev_center_counts, st));
            } else if (!p->center_counts_known && hipEventQuery(p->ev_center_counts) == hipSuccess) {
                p->center_counts[0] = p->h_center_counts[0];
                p->center_counts[1] = p->h_center_counts[1];
                p->center_counts_known = true;
            }
            // PC_CENTER_DEBUG: how long every dispatched wave ran (wall clock ticks), printed after the launch
            DevBuf<unsigned long long> d_dbg;
            const bool dbg_on = e->knobs.center_debug != 0 || e->want_center_steps;
            const size_t dbg_slots = (size_t)(kCenterCap * nchunks);   // heavy entries from the front, light ones from the back
            if (dbg_on) {
                rc = d_dbg.reserve(3 * dbg_slots);
                if (rc != PC_OK) return rc;
                HIP_TRY(hipMemsetAsync(d_dbg.p, 0, 3 * dbg_slots * 8, st));
            }
            unsigned long long *dbg = dbg_on ? d_dbg.p : nullptr;
            // (files with reads beyond a stream entry's 8-bit fields, or a stream too long for 32-bit byte offsets, take the
            // instantiation that tests every batch for them)
            bool general = false;
            for (auto *f : e->files) general |= f->len_max > 255 || f->n + f->nrun >= ((int64_t)1 << 28);
            {
                // descriptors: heavy entries one wave each, PC_CENTER_PER_WAVE light entries per wave (an eighth of the list per XCD)
                Center2Ctx c2;
                StagedFile *sf0 = e->files[0];
                c2.slots = p->d_cslots.p;
                c2.indirect = 0u;
                for (int k = 0; k < 3; ++k) {
                    c2.ent[k] = sf0->cs_n[k] >= 0 ? sf0->cs_ent[k].p : nullptr;
                    if (sf0->len_max > 255) c2.indirect |= 1u << k;
                }
                c2.files = e->d_files.p; c2.nfiles = nfiles; c2.file0 = sf0->view(); c2.mp = mp; c2.W = W; c2.inv = e->d_inv.p; c2.invh = e->d_invh.p; c2.cvalh = e->d_cvalh.p;
                c2.counters = p->d_ccounts.p;
                c2.known = p->center_counts_known ? 1u : 0u; c2.n_heavy = p->center_counts[0]; c2.n_light = p->center_counts[1];
                c2.opieces = p->d_opieces.p; c2.out = (double *)p->d_out.p; c2.norm_sum = e->norm_sum; c2.norm_on = e->norm_on ? 1 : 0;
                c2.dbg = dbg; c2.dbg_cap = (uint32_t)dbg_slots;
                uint64_t g2;
                if (p->center_counts_known) {
                    const uint64_t n8 = ((uint64_t)p->center_counts[1] + 7) >> 3;
                    g2 = (uint64_t)p->center_counts[0] + 8 * ((n8 + PC_CENTER_PER_WAVE - 1) / PC_CENTER_PER_WAVE);
                } else g2 = 2 * (uint64_t)nchunks + 8;
                const dim3 cg2((unsigned)std::max<uint64_t>(g2, 1));
                const bool multi = nfiles > 1;   // one descriptor per entry and file, replayed into the same sums in file order
                {
#define PC_LAUNCH_CENTER2(D, G, M) hipLaunchKernelGGL((k_center2<D, G, M>), cg2, dim3(64), (size_t)e->knobs.center_lds, st, c2)
#define PC_LAUNCH_CENTER2_M(D, G) do { if (multi) PC_LAUNCH_CENTER2(D, G, true); else PC_LAUNCH_CENTER2(D, G, false); } while (0)
                    if (dbg_on) { if (general) PC_LAUNCH_CENTER2_M(true, true); else PC_LAUNCH_CENTER2_M(true, false); }
                    else { if (general) PC_LAUNCH_CENTER2_M(false, true); else PC_LAUNCH_CENTER2_M(false, false); }
#undef PC_LAUNCH_CENTER2_M
#undef PC_LAUNCH_CENTER2
                }
            }
            if (dbg_on) {   // diagnostic launch: replay steps and dispatched waves, summed on the host; PC_CENTER_DEBUG prints them
                std::vector<unsigned long long> h(2 * dbg_slots), h_slots(dbg_slots);
                HIP_TRY(hipStreamSynchronize(st));
                HIP_TRY(hipMemcpy(h.data(), d_dbg.p, h.size() * 8, hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(h_slots.data(), d_dbg.p + h.size(), h_slots.size() * 8, hipMemcpyDeviceToHost));
                e->center_steps = 0; e->center_waves = 0;
                unsigned long long t0 = ~0ull, t1 = 0, sum = 0, steps_heavy = 0, steps_pers = 0, n_heavy_w = 0, n_pers_w = 0;
                const size_t n_front = (size_t)nchunks;   // (heavy entries occupy the front of the list: fewer than the chunk count)
                std::vector<std::pair<unsigned long long, size_t>> byd;
                for (size_t i = 0; i < dbg_slots; ++i)
                    if (h[2 * i]) {
                        e->center_steps += (int64_t)h_slots[i]; e->center_waves += 1;
                        const bool pers = i >= n_front;
                        (pers ? steps_pers : steps_heavy) += h_slots[i];
                        (pers ? n_pers_w : n_heavy_w) += 1;
                        t0 = std::min(t0, h[2 * i + 1]); t1 = std::max(t1, h[2 * i + 1] + h[2 * i]);
                        sum += h[2 * i];
                        byd.emplace_back(h[2 * i], i);
                    }
                if (e->knobs.center_debug != 0) {
                    fprintf(stderr, "[center] replay steps: %llu in %llu waves of the heavy entries, %llu in %llu waves of the light ones\n", steps_heavy, n_heavy_w,
                            steps_pers, n_pers_w);
                    fprintf(stderr, "[center] W %d: launch span %llu ticks (100 MHz: %.3f ms), summed wave time %llu ticks = %.1f x the span\n", W, t1 - t0,
                            (t1 - t0) / 1e5, sum, (double)sum / (double)std::max<unsigned long long>(t1 - t0, 1));
                    const int nb = 20;
                    std::vector<double> occ(nb, 0.0);
                    const double span = (double)std::max<unsigned long long>(t1 - t0, 1);
                    for (size_t i = 0; i < dbg_slots; ++i)
                        if (h[2 * i]) {
                            const double a = (double)(h[2 * i + 1] - t0) / span * nb, b = (double)(h[2 * i + 1] + h[2 * i] - t0) / span * nb;
                            for (int k = std::max(0, (int)a); k < nb && k < b; ++k) occ[(size_t)k] += std::min(b, k + 1.0) - std::max(a, (double)k);
                        }
                    fprintf(stderr, "[center] resident waves per twentieth of the launch:");
                    for (int k = 0; k < nb; ++k) fprintf(stderr, " %.0f", occ[(size_t)k]);
                    fprintf(stderr, "\n");
                    std::sort(byd.rbegin(), byd.rend());
                    for (size_t k = 0; k < std::min<size_t>(byd.size(), 8); ++k) {
                        const size_t i = byd[k].second;
                        fprintf(stderr, "[center]   slot %zu (%s): %.3f ms, started at %.3f ms, %llu steps\n", i,
                                i >= n_front ? "light" : "heavy", byd[k].first / 1e5, (h[2 * i + 1] - t0) / 1e5, h_slots[i]);
                    }
                    if (!byd.empty()) {
                        const size_t i = byd.back().second;
                        fprintf(stderr, "[center]   shortest: slot %zu: %.3f ms, %llu steps\n", i, byd.back().first / 1e5, h_slots[i]);
                    }
                }
            }
        }
        if (e->prof_level >= 1) HIP_TRY(hipEventRecord(e->ev[3], st));
        if (e->prof_level >= 2) HIP_TRY(hipEventRecord(e->ev[4], st));
    }
    if (e->prof_level >= 1) HIP_TRY(hipEventRecord(e->ev[5], st));
    HIP_TRY(hipGetLastError());
    p->last_dtype = out_dtype;
    p->counted = true;
    p->rle_runs = -1;
    e->timing_valid = e->prof_level > 0;
    e->timed_level = e->prof_level;
    // records once (8 B) + extra runs (8 B) + segments (24 B) + outputs once (8 B)
    e->last_alg_bytes = nrec * 8 + (nextra > 0 ? (nextra - 0) * 8 : 0) + p->nseg * 24 + p->covered * 8;
    return PC_OK;
}

int pc_sync(pc_engine *e) {
    if (!e) return fail(PC_ERR_ARG, "engine is NULL");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return PC_OK;
}


// Exact grids (pc_count) rest on the work counts of a plan being a function of (plan, work generation).  The last
// kernel of a count compares what was queued with what was launched; a mismatch means work items went unserved.
static int check_grid_guard(pc_engine *e, pc_plan *p) {
    if (!p->exact_grid_used && !p->guard_pending) return PC_OK;
    uint32_t err = 0;
    HIP_TRY(hipMemcpyAsync(&err, e->d_counters.p + 12, sizeof(err), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    p->guard_pending = false;
    if (!err) return PC_OK;
    HIP_TRY(hipMemsetAsync(e->d_counters.p + 12, 0, sizeof(uint32_t), e->stream));
    p->work_counts_known = false;
    p->work_counts_generation = 0;
    p->exact_grid_used = false;
    p->work_valid = false;
    if (err & 2u)
        return fail(PC_ERR_STATE, "the work lists of this plan overflowed their capacity (results incomplete): an engine defect -- please report "
                                  "the annotation / alignment shape; PC_WORK_R changes the item size");
    return fail(PC_ERR_STATE, "a count of this plan queued more work items than the cached work counts launched (results incomplete); "
                              "the cache has been dropped -- count again");
}

static constexpr size_t kSmallRead = 256 * 1024;

int pc_read_counts(pc_engine *e, pc_plan *p, void *host_out, int64_t out_elems) {
    if (!e || !p || p->e != e || !p->counted) return fail(PC_ERR_STATE, "pc_read_counts: nothing counted yet");
    if (out_elems != p->out_elems || (out_elems > 0 && !host_out)) return fail(PC_ERR_ARG, "pc_read_counts: buffer size mismatch");
    HIP_TRY(hipSetDevice(e->device));
    { const int grc = check_grid_guard(e, p); if (grc != PC_OK) return grc; }
    const size_t bytes = (size_t)out_elems * 8;
    const char *knob = getenv("PC_STAGE_SLICE");   // (test knob: the ring for every size, in pieces of so many 4 KiB pages)
    if (bytes >= 4 * TransferRing::kPiece || (bytes > 0 && knob)) {
        // the counts of a whole annotation: through the ring of page-locked pieces (a pageable destination the runtime has
        // not seen before is filled at 25 GB/s; see TransferRing)
        HIP_TRY(hipStreamSynchronize(e->stream));
        const std::vector<TransferJob> job{{host_out, p->d_out.p, bytes}};
        const size_t piece = knob ? (size_t)std::max<int64_t>(1, std::min<int64_t>(atoll(knob), 4096)) * 4096 : TransferRing::kPiece;
        return TransferRing::of(e->device).run(e->device, job, piece, knob != nullptr, true);
    }
    // (growing the buffer frees the old one: not while a plan upload may still be reading from it)
    if (bytes > e->pinned.cap && e->pinned_busy) { HIP_TRY(hipEventSynchronize(e->ev_pinned)); e->pinned_busy = false; }
    if (bytes > 0 && bytes <= kSmallRead && e->pinned.reserve(bytes) == PC_OK) {
        // short vectors come back through the page-locked buffer (stream order keeps it behind any
        // plan upload still reading from it): a pageable destination costs an extra staging hop
        HIP_TRY(hipMemcpyAsync(e->pinned.p, p->d_out.p, bytes, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        e->pinned_busy = false;
        memcpy(host_out, e->pinned.p, bytes);
        return PC_OK;
    }
    if (bytes > 0) HIP_TRY(hipMemcpyAsync(host_out, p->d_out.p, bytes, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return PC_OK;
}

int pc_plan_coordinates(pc_engine *e, pc_plan *p, int64_t *host_out, int64_t out_elems) {
    if (!e || !p || p->e != e) return fail(PC_ERR_ARG, "pc_plan_coordinates: bad engine/plan");
    if (out_elems != p->out_elems || (out_elems > 0 && !host_out)) return fail(PC_ERR_ARG, "pc_plan_coordinates: buffer size mismatch");
    HIP_TRY(hipSetDevice(e->device));
    if (out_elems == 0) return PC_OK;
    DevBuf<int64_t> d;
    d.pool = &e->pool;
    int rc = d.reserve((size_t)out_elems);
    if (rc == PC_OK) rc = ensure_gather_tables(e, p);   // the per-segment gather list of a large plan
    if (rc != PC_OK) return rc;
    hipStream_t st = e->stream;
    HIP_TRY(hipMemsetAsync(d.p, 0xff, (size_t)out_elems * 8, st));   // -1: elements no segment covers
    const unsigned grid = (unsigned)p->n_gchunks;
    if (grid) hipLaunchKernelGGL(k_coordinates, dim3(grid), dim3(kWG), 0, st, p->d_gsegs.p, p->d_gchunks.p, p->rows, d.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(host_out, d.p, (size_t)out_elems * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PC_OK;
}

void *pc_counts_device_ptr(pc_plan *p) { return p ? (void *)p->d_out.p : nullptr; }
void *pc_stream(pc_engine *e) { return e ? (void *)e->stream : nullptr; }
void *pc_total_device_ptr(pc_plan *p) { return p ? (void *)p->d_total.p : nullptr; }

int pc_total(pc_engine *e, pc_plan *p, void *host_out8) {
    if (!e || !p || p->e != e || !p->counted) return fail(PC_ERR_STATE, "pc_total: nothing counted yet");
    HIP_TRY(hipSetDevice(e->device));
    { const int grc = check_grid_guard(e, p); if (grc != PC_OK) return grc; }
    hipStream_t st = e->stream;
    HIP_TRY(hipMemsetAsync(p->d_total.p, 0, 8, st));
    if (p->out_elems > 0) {
        if (p->last_dtype == PC_OUT_INT64) {
            hipLaunchKernelGGL(k_total_i64, dim3(1024), dim3(kWG), 0, st, (const int64_t *)p->d_out.p, p->out_elems, (int64_t *)p->d_total.p);
        } else {
            const int nb = 1024;
            int rc = e->d_partial.reserve(nb);
            if (rc != PC_OK) return rc;
            hipLaunchKernelGGL(k_total_f64_partial, dim3(nb), dim3(kWG), 0, st, (const double *)p->d_out.p, p->out_elems, e->d_partial.p);
            hipLaunchKernelGGL(k_total_f64_final, dim3(1), dim3(64), 0, st, e->d_partial.p, nb, (double *)p->d_total.p);
        }
    }
    if (host_out8) HIP_TRY(hipMemcpyAsync(host_out8, p->d_total.p, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PC_OK;
}

// ------------------------------------------------------------------ export: run-length encoding
int pc_rle(pc_engine *e, pc_plan *p, int64_t period, int64_t *n_runs) {
    if (!e || !p || p->e != e || !n_runs) return fail(PC_ERR_ARG, "pc_rle: bad arguments");
    if (!p->counted) return fail(PC_ERR_STATE, "pc_rle: nothing counted yet");
    if (period < 0) return fail(PC_ERR_ARG, "pc_rle: period must be >= 0");
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t st = e->stream;
    const int64_t n = p->out_elems;
    p->rle_runs = 0;
    *n_runs = 0;
    if (n == 0) return PC_OK;
    const int64_t nwg = (n + kRleChunk - 1) / kRleChunk;
    int rc = p->d_rle_cnt.reserve((size_t)nwg);
    if (rc == PC_OK) rc = p->d_rle_base.reserve((size_t)nwg + 1);
    if (rc != PC_OK) return rc;
    const unsigned long long *v = (const unsigned long long *)p->d_out.p;
    hipLaunchKernelGGL(k_rle_count, dim3((unsigned)nwg), dim3(kWG), 0, st, v, n, period, p->d_rle_cnt.p);
    hipLaunchKernelGGL(k_rle_scan, dim3(1), dim3(kWG), 0, st, p->d_rle_cnt.p, nwg, p->d_rle_base.p, p->d_rle_base.p + nwg);
    int64_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, p->d_rle_base.p + nwg, sizeof(total), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    rc = p->d_rle_starts.reserve((size_t)std::max<int64_t>(total, 1));
    if (rc == PC_OK) rc = p->d_rle_values.reserve((size_t)std::max<int64_t>(total, 1));
    if (rc != PC_OK) return rc;
    hipLaunchKernelGGL(k_rle_write, dim3((unsigned)nwg), dim3(kWG), 0, st, v, n, period, p->d_rle_base.p, p->d_rle_starts.p,
                       p->d_rle_values.p);
    HIP_TRY(hipGetLastError());
    p->rle_runs = total;
    *n_runs = total;
    return PC_OK;
}

int pc_read_rle(pc_engine *e, pc_plan *p, int64_t *starts, void *values, int64_t n_runs) {
    if (!e || !p || p->e != e) return fail(PC_ERR_ARG, "pc_read_rle: bad arguments");
    if (p->rle_runs < 0) return fail(PC_ERR_STATE, "pc_read_rle: pc_rle has not run");
    if (n_runs != p->rle_runs || (n_runs > 0 && (!starts || !values))) return fail(PC_ERR_ARG, "pc_read_rle: expected %lld runs", (long long)p->rle_runs);
    HIP_TRY(hipSetDevice(e->device));
    if (n_runs > 0) {
        HIP_TRY(hipMemcpyAsync(starts, p->d_rle_starts.p, (size_t)n_runs * 8, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipMemcpyAsync(values, p->d_rle_values.p, (size_t)n_runs * 8, hipMemcpyDeviceToHost, e->stream));
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    return PC_OK;
}

int pc_set_profiling(pc_engine *e, int level) {
    if (!e) return fail(PC_ERR_ARG, "engine is NULL");
    if (level < 0 || level > 2) return fail(PC_ERR_ARG, "pc_set_profiling: level must be 0, 1 or 2");
    e->prof_level = level;
    e->timing_valid = false;
    return PC_OK;
}

int pc_last_timing(pc_engine *e, double *ms, int n) {
    if (!e || !ms || n <= 0) return fail(PC_ERR_ARG, "pc_last_timing: bad arguments");
    if (!e->timing_valid) return fail(PC_ERR_STATE, "pc_last_timing: no timed pc_count yet (see pc_set_profiling)");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipEventSynchronize(e->ev[5]));
    float t;
    for (double &v : e->last_ms) v = 0.0;
    HIP_TRY(hipEventElapsedTime(&t, e->ev[0], e->ev[5])); e->last_ms[0] = t;
    HIP_TRY(hipEventElapsedTime(&t, e->ev[2], e->ev[3])); e->last_ms[2] = t;
    if (e->timed_level >= 2) {
        HIP_TRY(hipEventElapsedTime(&t, e->ev[1], e->ev[2])); e->last_ms[1] = t;
        HIP_TRY(hipEventElapsedTime(&t, e->ev[3], e->ev[4])); e->last_ms[3] = t;
        HIP_TRY(hipEventElapsedTime(&t, e->ev[4], e->ev[5])); e->last_ms[4] = t;
        HIP_TRY(hipEventElapsedTime(&t, e->ev[0], e->ev[1])); e->last_ms[5] = t;
    }
    const int k = std::min(n, 6);
    for (int i = 0; i < k; ++i) ms[i] = e->last_ms[i];
    return k;
}

int64_t pc_last_algorithmic_bytes(pc_engine *e) { return e ? e->last_alg_bytes : -1; }

int pc_center_replay_steps(pc_engine *e, pc_plan *p, int64_t *steps, int64_t *waves) {
    if (!e || !p || p->e != e || !steps || !waves) return fail(PC_ERR_ARG, "pc_center_replay_steps: bad arguments");
    if (e->kind != PC_MAP_CENTER) return fail(PC_ERR_STATE, "pc_center_replay_steps: the mapping rule is not the center rule");
    e->want_center_steps = true;
    const int rc = pc_count(e, p, PC_OUT_FLOAT64);
    e->want_center_steps = false;
    if (rc != PC_OK) return rc;
    *steps = e->center_steps;
    *waves = e->center_waves;
    return PC_OK;
}

// ---- one segment in one call
// `ga[segment]` / `ga.get(segment)` (genome_array.py:861-928; the reference's scripts ask region by region, bin/psite.py:181-192):
// no plan object, no table upload, no read-back copy.  The window and its output piece travel in the kernel's arguments
// (k_hist_point<..., SINGLE> looks its record ranges up itself); the kernel writes the counts into page-locked host memory
// and a flag word behind them, which this call polls.  What a query paid for before was API calls and DMA hops, not
// bytes: plan create + upload + launch + read-back + sync, 59 us through the mirror at the end of round 4.
constexpr int kQueryMax = 4096;   // positions of one argument-borne window: 16 KiB of 32-bit bins

int pc_query_segment(pc_engine *e, int32_t tid, int64_t start, int64_t end, uint8_t strand, int reverse_out, int out_dtype, void *host_out) {
    if (!e || !host_out) return fail(PC_ERR_ARG, "pc_query_segment: bad arguments");
    if (!e->have_map) return fail(PC_ERR_STATE, "pc_query_segment: no mapping rule set (pc_set_mapping)");
    if (e->files.size() != 1) return fail(PC_ERR_STATE, "pc_query_segment: needs exactly one staged alignment file (several: pc_plan_create)");
    if (e->kind != PC_MAP_FIVE && e->kind != PC_MAP_THREE && e->kind != PC_MAP_VAR5)
        return fail(PC_ERR_STATE, "pc_query_segment: the center and the stratified rule go through pc_plan_create");
    if (out_dtype != PC_OUT_INT64 && out_dtype != PC_OUT_FLOAT64) return fail(PC_ERR_ARG, "pc_query_segment: bad out_dtype");
    if (e->norm_on && out_dtype != PC_OUT_FLOAT64) return fail(PC_ERR_ARG, "pc_query_segment: normalisation produces float64");
    const int64_t len = end - start;
    if (len <= 0 || len > kQueryMax || start < 0 || end > 0x7fffffffLL) return fail(PC_ERR_STATE, "pc_query_segment: the segment does not fit one window (1 .. %d positions)", kQueryMax);
    if (tid < 0 || tid >= e->ntid) return fail(PC_ERR_ARG, "pc_query_segment: reference id out of range");
    StagedFile *sf = e->files[0];
    if (e->ff_on && sf->n > 0 && !sf->have_sam) return fail(PC_ERR_STATE, "pc_query_segment: a FLAG / MAPQ filter is set but the alignment file has no FLAG / MAPQ columns");
    if (e->ff_max_nh && sf->n > 0 && !sf->have_nh) return fail(PC_ERR_STATE, "pc_query_segment: an NH filter is set but the alignment file has no NH column");
    HIP_TRY(hipSetDevice(e->device));
    if (!e->q_host) {
        HIP_TRY(hipHostMalloc((void **)&e->q_host, (size_t)kQueryMax * 8 + 64, hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer(&e->q_dev, e->q_host, 0));
        std::memset(e->q_host, 0, (size_t)kQueryMax * 8 + 64);
    }
    hipStream_t st = e->stream;
    const MapParams mp = e->params();
    int lmin = sf->len_min, lmax = sf->len_max;
    int tab_lo = 0, tab_n = 0;
    if (e->kind == PC_MAP_VAR5 && lmax >= lmin) {
        tab_lo = lmin;
        tab_n = std::max(0, std::min(std::min(lmax, e->table_len - 1) - lmin + 1, 1024));
    }
    int fast_lo = std::min(sf->tlen_min, sf->tlen_max), fast_hi = sf->tlen_max;
    const int G = kQueryMax;
    const size_t stage_words = (size_t)kOpStage * sizeof(OutPiece) / sizeof(uint32_t);
    const size_t lds = ((size_t)G + (size_t)((tab_n + 3) & ~3) + stage_words + (size_t)(fast_hi + 1) * kModes + 64) * sizeof(uint32_t);
    if (lds > e->max_lds) return fail(PC_ERR_STATE, "pc_query_segment: the window needs %zu bytes of LDS", lds);
    const int mode = mode_of(strand);
    Tile tl{};
    tl.tid = tid; tl.win_start = (int32_t)start; tl.piece_begin = tl.piece_end = 0; tl.mode_mask = 1u << mode;
    tl.op_begin = 0; tl.op_end = 1; tl.span_lo = 0; tl.span_hi = (uint16_t)len;
    OutPiece op{};
    op.out_off = reverse_out ? len - 1 : 0; op.row_stride = len; op.hist_off = 0; op.start = (int32_t)start; op.len = (int32_t)len;
    op.mode = mode; op.step = reverse_out ? -1 : 1;
    const uint32_t seq = ++e->q_seq ? e->q_seq : ++e->q_seq;   // (never 0: the flag's resting value)
    volatile uint32_t *flag = (volatile uint32_t *)(e->q_host + (size_t)kQueryMax * 8);
    uint32_t *d_flag = (uint32_t *)((uint8_t *)e->q_dev + (size_t)kQueryMax * 8);
    const FileView fv0 = sf->view();
    const int outmode = e->norm_on ? 2 : (out_dtype == PC_OUT_FLOAT64 ? 1 : 0);
#define PC_LAUNCH_QUERY(K, O)                                                                                          \
    hipLaunchKernelGGL((k_hist_point<K, O, kHistWG, false, false, true>), dim3(1), dim3(kHistWG), lds, st, (const Piece *)nullptr, (const OutPiece *)nullptr, \
                       fv0, fv0, (const FileView *)nullptr, (const WorkItem *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr, mp, G, 1,   \
                       tab_lo, tab_n, fast_lo, fast_hi, (uint32_t *)nullptr, (int64_t)e->Ws(), (OutT_<O>::type *)e->q_dev,                               \
                       e->norm_sum, (uint32_t)e->Wg(), (uint32_t)e->Wr(), (const FileRange *)nullptr, 1, tl, op, d_flag, seq)
#define PC_LAUNCH_QUERY_O(K)                                                                                           \
    do {                                                                                                               \
        if (outmode == 0) PC_LAUNCH_QUERY(K, 0);                                                                       \
        else if (outmode == 1) PC_LAUNCH_QUERY(K, 1);                                                                  \
        else PC_LAUNCH_QUERY(K, 2);                                                                                    \
    } while (0)
    switch (e->kind) {
    case PC_MAP_FIVE: PC_LAUNCH_QUERY_O(0); break;
    case PC_MAP_THREE: PC_LAUNCH_QUERY_O(1); break;
    default: PC_LAUNCH_QUERY_O(3); break;
    }
#undef PC_LAUNCH_QUERY_O
#undef PC_LAUNCH_QUERY
    HIP_TRY(hipGetLastError());
    // poll the flag the kernel writes behind its counts (a stream synchronisation costs more than the kernel runs);
    // fall back to the synchronisation if it does not show up soon
    const auto t0 = std::chrono::steady_clock::now();
    bool seen = false;
    for (uint64_t spin = 0;; ++spin) {
        if (*flag == seq) { seen = true; break; }
        if ((spin & 1023u) == 1023u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2e-3) break;
    }
    if (!seen) {
        HIP_TRY(hipStreamSynchronize(st));
        if (*flag != seq) return fail(PC_ERR_STATE, "pc_query_segment: the kernel did not report completion");
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    std::memcpy(host_out, e->q_host, (size_t)len * 8);
    return PC_OK;
}

int pc_center_row_fill(pc_engine *e, pc_plan *p, int64_t *row_entries, int64_t *row_slots) {
    if (!e || !p || p->e != e || !row_entries || !row_slots) return fail(PC_ERR_ARG, "pc_center_row_fill: bad arguments");
    if (!p->center_slots || p->center_generation != e->work_generation || !p->d_ccounts.p)
        return fail(PC_ERR_STATE, "pc_center_row_fill: the plan has no center dispatch list of one alignment file (count it under the center rule first)");
    HIP_TRY(hipSetDevice(e->device));
    unsigned long long v[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(v, p->d_ccounts.p + 4, sizeof(v), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    *row_entries = (int64_t)v[0];
    *row_slots = (int64_t)v[1];
    return PC_OK;
}

int pc_stream_probe(pc_engine *e, int64_t bytes, int iters, double *read_gbps, double *write_gbps) {
    if (!e || bytes < (1 << 20) || iters < 1) return fail(PC_ERR_ARG, "pc_stream_probe: bad arguments");
    HIP_TRY(hipSetDevice(e->device));
    DevBuf<uint8_t> buf;
    int rc = buf.reserve((size_t)bytes);
    if (rc != PC_OK) return rc;
    hipStream_t st = e->stream;
    const int64_t nvec = bytes / 16, n8 = bytes / 8;
    const unsigned grid_r = (unsigned)((nvec + kProbeChunk - 1) / kProbeChunk), grid_w = (unsigned)((n8 + 2 * kProbeChunk - 1) / (2 * kProbeChunk));
    uint32_t *sink = e->d_counters.p + 7;   // a word no kernel of the counting path reads
    float ms = 0.f;
    for (int pass = 0; pass < 2; ++pass) {   // pass 0: stores (also fills the buffer), pass 1: loads
        for (int it = -1; it < iters; ++it) {   // one untimed launch first
            if (it == 0) HIP_TRY(hipEventRecord(e->ev[6], st));
            if (pass == 0) hipLaunchKernelGGL(k_probe_write, dim3(grid_w), dim3(kWG), 0, st, (unsigned long long *)buf.p, n8);
            else hipLaunchKernelGGL(k_probe_read, dim3(grid_r), dim3(kWG), 0, st, (const u32x4 *)buf.p, nvec, sink);
        }
        HIP_TRY(hipEventRecord(e->ev[7], st));
        HIP_TRY(hipEventSynchronize(e->ev[7]));
        HIP_TRY(hipEventElapsedTime(&ms, e->ev[6], e->ev[7]));
        const double gbps = (double)bytes * iters / ((double)ms * 1e-3) / 1e9;
        if (pass == 0 && write_gbps) *write_gbps = gbps;
        if (pass == 1 && read_gbps) *read_gbps = gbps;
    }
    HIP_TRY(hipGetLastError());
    return PC_OK;
}

// ------------------------------------------------------------------ warnings
int pc_warn_flags(pc_engine *e, pc_plan *p, uint8_t *flags) { return pc_warn_details(e, p, flags, nullptr); }

int pc_warn_details(pc_engine *e, pc_plan *p, uint8_t *flags, int32_t *last_len) {
    if (!e || !p || p->e != e || (p->nseg > 0 && !flags)) return fail(PC_ERR_ARG, "pc_warn_flags: bad arguments");
    if (!e->have_map) return fail(PC_ERR_STATE, "pc_warn_flags: no mapping rule set");
    std::memset(flags, 0, (size_t)p->nseg);
    if (last_len) std::fill(last_len, last_len + p->nseg, (int32_t)-1);
    if (e->kind == PC_MAP_STRAT5) return PC_OK; // never warns
    // cheap pre-check on the per-length record histogram (ignores filters: conservative)
    bool any = false;
    for (auto *f : e->files) {
        for (int L = f->len_min; L <= f->len_max && !any; ++L) {   // lengths present in the file
            if (!f->len_hist[(size_t)L]) continue;
            bool bad;
            switch (e->kind) {
            case PC_MAP_FIVE: case PC_MAP_THREE: bad = e->param >= L; break;
            case PC_MAP_CENTER: bad = L - 2 * e->param < 0; break;
            default: bad = L >= e->table_len || e->h_fw[(size_t)L] < 0;
            }
            any |= bad;
        }
    }
    if (!any) return PC_OK;
    HIP_TRY(hipSetDevice(e->device));
    int rc = refresh_file_views(e);
    if (rc != PC_OK) return rc;
    const MapParams mp = e->params();
    std::vector<Unmappable> all;
    std::vector<uint32_t> file_of;   // staged file of every entry of `all` (parallel array, permuted with it)
    int file_index = -1;
    for (auto *f : e->files) {
        ++file_index;
        if (!f->n) continue;
        uint32_t cap = 1u << 16;
        for (;;) {
            rc = e->d_unmap.reserve(cap);
            if (rc != PC_OK) return rc;
            HIP_TRY(hipMemsetAsync(e->d_counters.p + 1, 0, sizeof(uint32_t), e->stream));
            hipLaunchKernelGGL(k_unmappable, dim3((unsigned)((f->n + kWG - 1) / kWG)), dim3(kWG), 0, e->stream, f->view(), mp, e->ntid,
                               e->d_unmap.p, cap, e->d_counters.p + 1);
            uint32_t cnt = 0;
            HIP_TRY(hipMemcpyAsync(&cnt, e->d_counters.p + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(hipStreamSynchronize(e->stream));
            if (cnt <= cap) {
                const size_t base = all.size();
                all.resize(base + cnt);
                if (cnt) HIP_TRY(hipMemcpy(all.data() + base, e->d_unmap.p, (size_t)cnt * sizeof(Unmappable), hipMemcpyDeviceToHost));
                file_of.resize(base + cnt, (uint32_t)file_index);
                break;
            }
            cap = cnt;
        }
    }
    if (all.empty()) return PC_OK;
    // host: does any unmappable record of the right strand overlap the segment (htslib rule)?
    {   // sort by (tid, pos), carrying the file index along
        std::vector<size_t> perm(all.size());
        for (size_t i = 0; i < perm.size(); ++i) perm[i] = i;
        std::sort(perm.begin(), perm.end(), [&](size_t a, size_t b) {
            if (all[a].tid != all[b].tid) return all[a].tid < all[b].tid;
            return all[a].pos < all[b].pos;
        });
        std::vector<Unmappable> a2(all.size());
        std::vector<uint32_t> f2(all.size());
        for (size_t i = 0; i < perm.size(); ++i) { a2[i] = all[perm[i]]; f2[i] = file_of[perm[i]]; }
        all.swap(a2);
        file_of.swap(f2);
    }
    const size_t n = all.size();
    std::vector<int32_t> pmax_f(n), pmax_r(n);
    for (size_t i = 0; i < n; ++i) {
        const bool fresh = i == 0 || all[i].tid != all[i - 1].tid;
        int32_t pf = fresh ? INT32_MIN : pmax_f[i - 1], pr = fresh ? INT32_MIN : pmax_r[i - 1];
        if (all[i].rev) pr = std::max(pr, all[i].end); else pf = std::max(pf, all[i].end);
        pmax_f[i] = pf;
        pmax_r[i] = pr;
    }
    { const int frc = fetch_host_inputs(p); if (frc != PC_OK) return frc; }
    for (int64_t s = 0; s < p->nseg; ++s) {
        const int32_t t = p->h_tid[(size_t)s];
        if (t < 0 || t >= e->ntid) continue;
        const int64_t st = p->h_start[(size_t)s], en = p->h_end[(size_t)s];
        // last record of tid t with pos < en
        size_t lo = 0, hi = n;
        while (lo < hi) {
            size_t mid = (lo + hi) / 2;
            if (all[mid].tid < t || (all[mid].tid == t && (int64_t)all[mid].pos < en)) lo = mid + 1; else hi = mid;
        }
        if (lo == 0 || all[lo - 1].tid != t) continue;
        const int mode = mode_of(p->h_strand[(size_t)s]);
        int64_t best;
        if (mode == 0) best = pmax_f[lo - 1];
        else if (mode == 1) best = pmax_r[lo - 1];
        else best = std::max(pmax_f[lo - 1], pmax_r[lo - 1]);
        if (best > st) flags[s] = 1;
        if (best > st && last_len) {
            // the LAST offending read in fetch order (file-major, then record order) among those fetch
            // returns for the segment: walk back while some earlier read of the strand can still reach it
            uint64_t best_key = 0;
            bool have = false;
            for (size_t i = lo; i-- > 0 && all[i].tid == t;) {
                const int64_t reach = mode == 0 ? pmax_f[i] : (mode == 1 ? pmax_r[i] : std::max(pmax_f[i], pmax_r[i]));
                if (reach <= st) break;
                const bool strand_ok = mode == 0 ? !all[i].rev : (mode == 1 ? all[i].rev != 0 : true);
                if (!strand_ok || (int64_t)all[i].end <= st) continue;
                const uint64_t key = ((uint64_t)file_of[i] << 32) | all[i].rec;
                if (!have || key > best_key) { best_key = key; last_len[s] = all[i].len; have = true; }
            }
        }
    }
    return PC_OK;
}

int pc_mapped_reads(pc_engine *e, int file, int64_t rec_lo, int64_t rec_hi, int32_t tid, int64_t start, int64_t end,
                    uint8_t strand, uint8_t *mask) {
    if (!e || file < 0 || file >= (int)e->files.size()) return fail(PC_ERR_ARG, "pc_mapped_reads: bad file index");
    if (!e->have_map) return fail(PC_ERR_STATE, "pc_mapped_reads: no mapping rule set");
    StagedFile *f = e->files[file];
    if (rec_lo < 0 || rec_hi > f->n || rec_hi < rec_lo || (rec_hi > rec_lo && !mask)) return fail(PC_ERR_ARG, "pc_mapped_reads: bad record range");
    (void)tid;
    if (rec_hi == rec_lo) return PC_OK;
    HIP_TRY(hipSetDevice(e->device));
    const int64_t n = rec_hi - rec_lo;
    DevBuf<uint8_t> d_mask;
    int rc = d_mask.reserve((size_t)n);
    if (rc != PC_OK) return rc;
    hipLaunchKernelGGL(k_mapped_reads, dim3((unsigned)((n + kWG - 1) / kWG)), dim3(kWG), 0, e->stream, f->view(), e->params(), rec_lo, rec_hi,
                       start, end, mode_of(strand), !(strand & PC_STRAND_NOFILTER), d_mask.p);
    HIP_TRY(hipMemcpyAsync(mask, d_mask.p, (size_t)n, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return PC_OK;
}

int pc_mapped_reads_batch(pc_engine *e, pc_plan *p, int64_t *offsets, int64_t *total) {
    if (!e || !p || p->e != e || !offsets || !total) return fail(PC_ERR_ARG, "pc_mapped_reads_batch: bad arguments");
    if (!e->have_map) return fail(PC_ERR_STATE, "pc_mapped_reads_batch: no mapping rule set");
    if (e->files.empty()) return fail(PC_ERR_STATE, "pc_mapped_reads_batch: no alignments staged");
    HIP_TRY(hipSetDevice(e->device));
    int rc = refresh_file_views(e);
    if (rc != PC_OK) return rc;
    const int nfiles = (int)e->files.size();
    const int64_t nseg = p->nseg, npair = nseg * nfiles;
    *total = 0;
    offsets[0] = 0;
    p->mr_total = 0;
    if (npair == 0) return PC_OK;
    if (npair >= (int64_t)0x7fffffff) return fail(PC_ERR_ARG, "pc_mapped_reads_batch: too many (segment, file) pairs");
    { const int frc = fetch_host_inputs(p); if (frc != PC_OK) return frc; }
    std::vector<BatchSeg> segs((size_t)nseg);
    for (int64_t s = 0; s < nseg; ++s) {
        BatchSeg &g = segs[(size_t)s];
        const int32_t t = p->h_tid[(size_t)s];
        g.tid = (t >= 0 && t < e->ntid) ? t : -1;
        g.start = std::max<int64_t>(p->h_start[(size_t)s], 0);
        g.end = std::min<int64_t>(p->h_end[(size_t)s], 0x7fffffffLL);
        g.mode = mode_of(p->h_strand[(size_t)s]) | ((p->h_strand[(size_t)s] & PC_STRAND_NOFILTER) ? 0x100 : 0);
    }
    std::vector<int64_t> spans;
    for (auto *f : e->files) spans.push_back(f->max_span);
    DevBuf<BatchSeg> d_segs;
    DevBuf<int64_t> d_spans;
    d_segs.pool = &e->pool; d_spans.pool = &e->pool; p->d_mr_off.pool = &e->pool; p->d_mr_rec.pool = &e->pool;
    hipStream_t st = e->stream;
    rc = d_segs.upload(segs, st);
    if (rc == PC_OK) rc = d_spans.upload(spans, st);
    if (rc == PC_OK) rc = p->d_mr_off.reserve((size_t)npair + 1);
    if (rc != PC_OK) return rc;
    const MapParams mp = e->params();
    HIP_TRY(hipMemsetAsync(p->d_mr_off.p + npair, 0, 8, st));
    hipLaunchKernelGGL((k_mapped_reads_batch<false>), dim3((unsigned)npair), dim3(kWG), 0, st, d_segs.p, nseg, e->d_files.p, nfiles, d_spans.p, mp,
                       p->d_mr_off.p, (const unsigned long long *)nullptr, (uint32_t *)nullptr);
    {
        size_t tmp_bytes = 0;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, p->d_mr_off.p, p->d_mr_off.p, (int)(npair + 1), st));
        DevBuf<uint8_t> d_tmp;
        d_tmp.pool = &e->pool;
        rc = d_tmp.reserve(std::max<size_t>(tmp_bytes, 16));
        if (rc != PC_OK) return rc;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tmp_bytes, p->d_mr_off.p, p->d_mr_off.p, (int)(npair + 1), st));
        static_assert(sizeof(unsigned long long) == sizeof(int64_t), "offsets are copied as they are");
        HIP_TRY(hipMemcpyAsync(offsets, p->d_mr_off.p, (size_t)(npair + 1) * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    const int64_t tot = offsets[npair];
    if (tot >= (int64_t)0xffffffffu) return fail(PC_ERR_ARG, "pc_mapped_reads_batch: more than 2^32-2 mapped reads in one batch; split the segments");
    rc = p->d_mr_rec.reserve((size_t)std::max<int64_t>(tot, 1));
    if (rc != PC_OK) return rc;
    hipLaunchKernelGGL((k_mapped_reads_batch<true>), dim3((unsigned)npair), dim3(kWG), 0, st, d_segs.p, nseg, e->d_files.p, nfiles, d_spans.p, mp,
                       (unsigned long long *)nullptr, p->d_mr_off.p, p->d_mr_rec.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));   // (the segment table and the spans return to the pool)
    p->mr_total = tot;
    *total = tot;
    return PC_OK;
}

int pc_read_mapped_reads(pc_engine *e, pc_plan *p, uint32_t *rec, int64_t total) {
    if (!e || !p || p->e != e) return fail(PC_ERR_ARG, "pc_read_mapped_reads: bad arguments");
    if (total != p->mr_total || (total > 0 && !rec)) return fail(PC_ERR_ARG, "pc_read_mapped_reads: expected %lld records (pc_mapped_reads_batch)", (long long)p->mr_total);
    HIP_TRY(hipSetDevice(e->device));
    if (total > 0) HIP_TRY(hipMemcpyAsync(rec, p->d_mr_rec.p, (size_t)total * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return PC_OK;
}

} // extern "C"

// ================================================================== compressed BAM on the GPU (bam_kernels.hip.h)
#include "bam_kernels.hip.h"

struct pc_bam {
    pc_engine *e = nullptr;
    std::string name;
    int64_t n = 0, nrun = 0, mapped = 0, unplaced = 0, total = 0;
    std::vector<std::string> ref_names;
    std::vector<int32_t> ref_lengths;
    DevBuf<int32_t> tid, pos, blk_start, blk_len;
    DevBuf<uint16_t> alen;
    DevBuf<uint8_t> flags, nblk;
    DevBuf<uint16_t> flag16;           // the SAM FLAG word, MAPQ and l_seq of every staged record (pc_bam_read_sam; the
    DevBuf<uint8_t> mapq;              // first two stay with the staged file for the FLAG / MAPQ filter)
    DevBuf<int32_t> lseq;
    DevBuf<uint16_t> nh;               // the NH:i tag of every staged record, 0 without one (pc_bam_read_nh; stays with the staged file for the NH filter)
    std::vector<int64_t> wide_idx;
    std::vector<int32_t> wide_alen, wide_nblk;
    double ms[4] = {0, 0, 0, 0};     // upload, inflate (+ CRC), record chain, fields + columns
    int64_t members = 0, inflated_bytes = 0, compressed_bytes = 0;
    int chain_restarts = 0;
};

namespace {

uint16_t brd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t brd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// CRC-32 tables (RFC 1952): the byte table, and the operator that advances the register over kCrcSlice zero bytes
// split by register byte (k_bgzf_crc combines 64 slice remainders with it)
struct CrcTables {
    uint32_t tab[256];
    uint32_t shift[4 * 256];
    CrcTables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
            tab[i] = c;
        }
        for (int b = 0; b < 4; ++b)
            for (uint32_t v = 0; v < 256; ++v) {
                uint32_t c = v << (8 * b);
                for (int k = 0; k < pcbam::kCrcSlice; ++k) c = tab[c & 0xffu] ^ (c >> 8);
                shift[b * 256 + v] = c;
            }
    }
};
const CrcTables &crc_tables() { static const CrcTables t; return t; }

double ms_between(hipEvent_t a, hipEvent_t b) { float t = 0.f; return hipEventElapsedTime(&t, a, b) == hipSuccess ? (double)t : 0.0; }

} // namespace

extern "C" {

int pc_bam_close(pc_bam *b) {
    if (!b) return PC_OK;
    if (b->e) { (void)hipSetDevice(b->e->device); (void)hipStreamSynchronize(b->e->stream); }
    delete b;
    return PC_OK;
}

namespace {
struct BamClock {   // PC_BAM_TIMING=1: wall-clock laps of the host side of the GPU decoder
    bool on = getenv("PC_BAM_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[bam] %-34s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
    void note(const char *what) {   // time since the last lap, the lap goes on
        if (!on) return;
        fprintf(stderr, "[bam]   (%s: %.2f ms into the lap)\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count());
    }
};
} // namespace

// `uploaded` (optional): called once, with the stream the image is uploaded on, when the last piece has been queued -- once
// that stream has drained the host copy of the file is not read again (pc_bam_open_path takes its mapping down while the
// GPU is still inflating).
typedef std::function<void(hipStream_t)> UploadedHook;
// A region read (pc_bam_open_span): only the BGZF members between two virtual offsets of the BAI index are uploaded and
// inflated (plus the leading members that hold the header), and only the records that overlap one of the regions stay.
struct BamSpan {
    uint64_t voff_begin = 0, voff_end = 0;   // (file offset of a member << 16 | offset in its payload): [begin, end); 0, 0: header only
    int nreg = 0;                            // merged regions, ascending by (reference id, start)
    const int32_t *tid = nullptr;
    const int64_t *beg = nullptr, *end = nullptr;
    int64_t header_bytes = (int64_t)256 << 10;   // compressed bytes from the start of the file searched for the header (grown on retry)
};
constexpr int PC_RETRY_HEADER = -1000;   // (internal) the header did not fit the leading members that were inflated
static int bam_open_impl(pc_engine *e, const void *image_, int64_t size, const char *name, pc_bam **out, const UploadedHook *uploaded, const BamSpan *span = nullptr);

int pc_bam_open(pc_engine *e, const void *image, int64_t size, const char *name, pc_bam **out) {
    return bam_open_impl(e, image, size, name, out, nullptr);
}

static int bam_open_impl(pc_engine *e, const void *image_, int64_t size, const char *name, pc_bam **out, const UploadedHook *uploaded, const BamSpan *span) {
    using namespace pcbam;
    if (!e || !out || size < 0 || (size > 0 && !image_)) return fail(PC_ERR_ARG, "pc_bam_open: bad arguments");
    *out = nullptr;
    const uint8_t *image = (const uint8_t *)image_;
    const std::string path = name ? name : "<memory>";
    HIP_TRY(hipSetDevice(e->device));
    PoolScope pool_scope(&e->pool);   // (the decoder's scratch -- image, inflated stream, record table -- is recycled through the engine's pool)
    hipStream_t st = e->stream;
    BamClock clk;
    // ---- member boundaries (host: a walk over the gzip headers; 18 + bytes per 64 KiB of payload)
    // one member at `off`: 0, or which defect (the messages below)
    auto parse_member = [&](int64_t off, Member &mb, int64_t &clen_out) -> int {
        if (off + 18 > size) return 1;
        const uint8_t *h = image + off;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return 2;
        const uint16_t xlen = brd16(h + 10);
        if (off + 12 + xlen > size) return 3;
        int bsize = -1;
        for (size_t x = 0; x + 4 <= xlen;) {
            const uint8_t *sf = h + 12 + x;
            const uint16_t slen = brd16(sf + 2);
            if (sf[0] == 'B' && sf[1] == 'C' && slen == 2 && x + 6 <= xlen) bsize = brd16(sf + 4);
            x += 4 + slen;
        }
        if (bsize < 0) return 4;
        const int64_t clen = (int64_t)bsize + 1;
        if (off + clen > size) return 5;
        const uint32_t isize = brd32(image + off + clen - 4);
        if (isize > (1u << 16)) return 6;
        const int64_t hdr = 12 + xlen;
        if (clen < hdr + 8) return 7;
        mb.coff = (uint64_t)(off + hdr); mb.clen = (uint32_t)(clen - hdr - 8); mb.ulen = isize; mb.uoff = 0;
        mb.crc = brd32(image + off + clen - 8); mb.pad = 0;
        clen_out = clen;
        return 0;
    };
    auto walk_error = [&](int code) -> int {
        switch (code) {
        case 1: return fail(PC_ERR_ARG, "truncated BGZF header");
        case 2: return fail(PC_ERR_ARG, "not a BGZF file (bad gzip member header)");
        case 3: return fail(PC_ERR_ARG, "truncated BGZF extra field");
        case 4: return fail(PC_ERR_ARG, "BGZF member without BC subfield");
        case 5: return fail(PC_ERR_ARG, "truncated BGZF member");
        case 6: return fail(PC_ERR_ARG, "corrupt BGZF member (more than 64 KiB of payload)");
        default: return fail(PC_ERR_ARG, "BGZF inflate failed in %s", path.c_str());
        }
    };
    std::vector<Member> members;
    // the parts of the file that go to the GPU: [file_lo, file_hi) lands at image offset dev_lo (one run: the whole file)
    struct Run { int64_t file_lo, file_hi, dev_lo; int m0, m1; };
    std::vector<Run> runs;
    int span_first_member = -1, span_last_member = -1;   // region read: indices (in `members`) of the members at voff_begin >> 16 and at voff_end >> 16
    if (span) {
        const int64_t cb = (int64_t)(span->voff_begin >> 16), ce = (int64_t)(span->voff_end >> 16);
        const bool have = span->voff_end > span->voff_begin;
        if (have && (cb >= size || ce > size)) return fail(PC_ERR_ARG, "the index does not belong to this BAM file (a chunk lies beyond its end): %s", path.c_str());
        // serial walks: the header's members from the start of the file, the span's from its first member on
        // (members that start in [lo, hi_excl), and the one at `last` too if asked for)
        auto walk_run = [&](int64_t lo, int64_t hi_excl, bool with_last, int64_t last, int *idx_last) -> int {
            Run r; r.file_lo = lo; r.m0 = (int)members.size(); r.dev_lo = 0;
            int64_t off = lo;
            while (off < size && (off < hi_excl || (with_last && off <= last))) {
                Member mb;
                int64_t clen = 0;
                const int code = parse_member(off, mb, clen);
                if (code) return walk_error(code);
                if (idx_last && off == last) *idx_last = mb.ulen ? (int)members.size() : -1;
                if (mb.ulen) members.push_back(mb);
                off += clen;
            }
            r.file_hi = off; r.m1 = (int)members.size();
            if (r.file_hi > r.file_lo) runs.push_back(r);
            return PC_OK;
        };
        int rc0 = walk_run(0, have ? std::min<int64_t>(span->header_bytes, cb) : span->header_bytes, false, 0, nullptr);
        if (rc0 != PC_OK) return rc0;
        if (have) {
            if (!runs.empty() && runs.back().file_hi > cb)   // (member starts are what the walk lands on: cb is none)
                return fail(PC_ERR_ARG, "the index does not belong to this BAM file (a chunk does not start at a BGZF member): %s", path.c_str());
            span_first_member = (int)members.size();
            rc0 = walk_run(cb, ce, (span->voff_end & 0xffffu) != 0, ce, &span_last_member);
            if (rc0 != PC_OK) return rc0;
        }
        // adjacent runs become one (a run is uploaded as one contiguous copy)
        for (size_t k = 1; k < runs.size();)
            if (runs[k].file_lo == runs[k - 1].file_hi) { runs[k - 1].file_hi = runs[k].file_hi; runs[k - 1].m1 = runs[k].m1; runs.erase(runs.begin() + (long)k); }
            else ++k;
        int64_t dev = 0;
        for (Run &r : runs) {   // the members' streams by their place in the image on the device
            r.dev_lo = dev;
            for (int m = r.m0; m < r.m1; ++m) members[(size_t)m].coff = (uint64_t)((int64_t)members[(size_t)m].coff - r.file_lo + r.dev_lo);
            dev += r.file_hi - r.file_lo;
        }
    }
    int64_t walked_to = span ? size : 0;
    // Large files: the walk is a chain of dependent cache misses (40 k members: 5.6 ms), so every host thread walks its
    // own stretch of the file from the first offset in it where three members in a row parse; a stretch counts only if
    // the walk of the stretch before it LANDS on its first member -- whatever does not chain is walked again, serially.
    const int64_t walk_min = getenv("PC_BAM_WALK_MIN") ? atoll(getenv("PC_BAM_WALK_MIN")) : ((int64_t)32 << 20);   // (tests: the parallel walk on small files)
    const int WT = !span && size >= walk_min && size >= 64 ? std::max(1, std::min(usable_cpus(), 16)) : 1;
    if (WT > 1) {
        struct Stretch { int64_t first = -1, landing = -1; std::vector<Member> mem; };
        std::vector<Stretch> str((size_t)WT);
        parallel_chunks((int64_t)WT, WT, [&](int, int64_t kb, int64_t ke) {
            for (int64_t k = kb; k < ke; ++k) {
                Stretch &sx = str[(size_t)k];
                const int64_t lo = size * k / WT, hi = size * (k + 1) / WT;
                int64_t off = lo;
                if (k > 0) {   // the first offset from which three members parse
                    off = -1;
                    for (int64_t c = lo; c < hi && c + 18 <= size; ++c) {
                        if (image[c] != 31 || image[c + 1] != 139) continue;
                        int64_t q = c;
                        bool ok = true;
                        for (int r = 0; r < 3 && ok && q < size; ++r) {
                            Member mb;
                            int64_t cl = 0;
                            ok = parse_member(q, mb, cl) == 0;
                            q += cl;
                        }
                        if (ok) { off = c; break; }
                    }
                    if (off < 0) continue;
                }
                sx.first = off;
                while (off < hi && off < size) {
                    Member mb;
                    int64_t cl = 0;
                    if (parse_member(off, mb, cl) != 0) { sx.first = -1; break; }   // (a defect: the serial walk below reports it)
                    if (mb.ulen) sx.mem.push_back(mb);
                    off += cl;
                }
                sx.landing = off;
            }
        });
        int64_t expected = 0;
        for (int k = 0; k < WT; ++k) {
            const Stretch &sx = str[(size_t)k];
            if (sx.first < 0 || sx.first != expected) break;
            members.insert(members.end(), sx.mem.begin(), sx.mem.end());
            expected = sx.landing;
        }
        walked_to = expected;
    }
    for (int64_t off = walked_to; off < size;) {
        Member mb;
        int64_t clen = 0;
        const int code = parse_member(off, mb, clen);
        if (code) return walk_error(code);
        if (mb.ulen) members.push_back(mb);      // (empty members -- the end-of-file marker -- hold nothing)
        off += clen;
    }
    uint64_t total_u = 0;
    for (Member &mb : members) { mb.uoff = total_u; total_u += mb.ulen; }
    if (!span) runs.push_back(Run{0, size, 0, 0, (int)members.size()});
    int64_t image_bytes = 0;
    for (const Run &r : runs) image_bytes += r.file_hi - r.file_lo;
    pc_bam *b = new pc_bam();
    clk.lap("member walk");
    b->e = e; b->name = path; b->members = (int64_t)members.size(); b->inflated_bytes = (int64_t)total_u; b->compressed_bytes = size;
    struct Guard { pc_bam *b; ~Guard() { if (b) pc_bam_close(b); } } guard{b};
    const int nm = (int)members.size();
    hipEvent_t ev[5];
    for (auto &x : ev) HIP_TRY(hipEventCreate(&x));
    struct EvGuard { hipEvent_t *ev; ~EvGuard() { for (int i = 0; i < 5; ++i) (void)hipEventDestroy(ev[i]); } } evg{ev};
    DevBuf<uint8_t> d_image, d_stream;
    DevBuf<Member> d_members;
    DevBuf<uint32_t> d_status, d_crc;
    int rc = d_image.reserve((size_t)std::max<int64_t>(image_bytes, 16) + 16);
    if (rc == PC_OK) rc = d_stream.reserve((size_t)total_u + 64);
    if (rc == PC_OK) rc = d_members.reserve((size_t)std::max(nm, 1));
    if (rc == PC_OK) rc = d_status.reserve((size_t)std::max(nm, 1));
    if (rc == PC_OK) rc = d_crc.reserve(5 * 256);
    if (rc != PC_OK) return rc;
    // An early return between here and the synchronisation behind the inflate launches must not hand the image, the
    // stream buffer or the page-locked ring back (nor let the caller unmap the file) while the side / auxiliary streams
    // still use them: drain every stream the decoder queues on before the buffers above go out of scope.
    struct Drain {
        pc_engine *e; bool armed;
        ~Drain() {
            if (!armed) return;
            if (e->side_stream) (void)hipStreamSynchronize(e->side_stream);
            for (int k = 0; k < pc_engine::kAux; ++k) if (e->aux_stream[k]) (void)hipStreamSynchronize(e->aux_stream[k]);
            (void)hipStreamSynchronize(e->stream);
        }
    } drain{e, true};
    clk.lap("allocations (image, stream)");
    HIP_TRY(hipEventRecord(ev[0], st));
    if (nm) HIP_TRY(hipMemcpyAsync(d_members.p, members.data(), (size_t)nm * sizeof(Member), hipMemcpyHostToDevice, st));
    const CrcTables &ct = crc_tables();
    HIP_TRY(hipMemcpyAsync(d_crc.p, ct.tab, sizeof(ct.tab), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_crc.p + 256, ct.shift, sizeof(ct.shift), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(d_stream.p + total_u, 0, 64, st));
    HIP_TRY(hipEventRecord(ev[1], st));
    // ---- upload + inflate, piece by piece: the file image crosses PCIe on the side stream in pieces of ~128 MiB of
    // whole members while the members of the pieces before are inflated on the main one (one wave per member).  (Every
    // launch ends in a tail of half-empty CUs -- a member takes ~4 ms and ~3 000 are in flight -- so the pieces are
    // large: 20 M aligner-like records, 578 MB: one piece 87 ms, 48 MiB pieces 67 ms, 128 MiB 58 ms, 256 MiB 61 ms.)
    std::vector<uint32_t> status((size_t)nm, 0u);
    if (nm) {
        const bool serial_symbols = getenv("PC_BGZF_SERIAL") != nullptr && atoi(getenv("PC_BGZF_SERIAL")) != 0;   // (round 4's first kernel, for comparison)
        const int64_t piece_bytes = getenv("PC_BAM_PIECE") ? std::max<int64_t>(1, atoll(getenv("PC_BAM_PIECE"))) : ((int64_t)64 << 20);
        hipStream_t up = e->side_stream ? e->side_stream : st;
        std::vector<hipEvent_t> landed;
        struct EvList { std::vector<hipEvent_t> &v; ~EvList() { for (auto x : v) (void)hipEventDestroy(x); } } landed_guard{landed};
        if (up != st) {   // the side stream starts behind what the main one has queued so far (the buffers' previous users)
            hipEvent_t x;
            HIP_TRY(hipEventCreateWithFlags(&x, hipEventDisableTiming));
            landed.push_back(x);
            HIP_TRY(hipEventRecord(x, st));
            HIP_TRY(hipStreamWaitEvent(up, x, 0));
        }
        // The inflate launches alternate between the main stream and an auxiliary one: a launch ends in a tail of
        // half-empty CUs (a member takes ~4 ms, ~3 000 are in flight), which the launch of the next piece fills.
        // (two streams in turn: measured on two boxes, 64 MiB pieces, 20 M aligner-like records: one stream 56 - 58 ms, two
        // 46.6 - 53.5, four 48.6; PC_BAM_STREAMS = 1 .. 4 for experiments)
        int naux = up != st ? 1 : 0;
        if (const char *env = getenv("PC_BAM_STREAMS")) naux = up != st ? std::max(0, std::min(pc_engine::kAux, atoi(env) - 1)) : 0;
        for (int k = 0; k < naux; ++k)   // (behind what the main stream has queued: the members table, the previous users of the buffers)
            HIP_TRY(hipStreamWaitEvent(e->aux_stream[k], landed[0], 0));
        // large files cross PCIe through two page-locked halves of one piece each (made once per engine)
        bool ring = up != st && image_bytes >= 2 * piece_bytes && !getenv("PC_BAM_NO_RING");
        bool ring_busy[2] = {false, false};
        const int ring_threads = std::max(1, std::min(usable_cpus(), 16));
        if (ring) {
            // (a piece ends with a whole member: up to 64 KiB beyond piece_bytes)
            for (int k = 0; k < 2 && ring; ++k) {
                if (e->bam_ring[k].reserve((size_t)piece_bytes + ((size_t)1 << 17)) != PC_OK) ring = false;
                if (ring && !e->ev_ring[k] && hipEventCreateWithFlags(&e->ev_ring[k], hipEventDisableTiming) != hipSuccess) ring = false;
            }
            (void)hipGetLastError();
        }
        int piece_no = 0;
        for (const Run &run : runs) {
        // (offsets in the image on the device; the file's bytes of the run start at run.file_lo - run.dev_lo before them)
        const uint8_t *run_src = image + (run.file_lo - run.dev_lo);
        int64_t byte0 = run.dev_lo;          // the run is uploaded from here on (gzip headers and trailers ride along)
        for (int m0 = run.m0; m0 < run.m1; ++piece_no) {
            int m1 = m0;
            int64_t byte1 = byte0;
            while (m1 < run.m1 && (byte1 - byte0 < piece_bytes || m1 == m0)) {
                byte1 = (int64_t)(members[(size_t)m1].coff + members[(size_t)m1].clen);
                ++m1;
            }
            if (m1 == run.m1) byte1 = run.dev_lo + (run.file_hi - run.file_lo);
            if (ring) {
                // through a page-locked half: the runtime's own staging of a pageable copy runs on one thread (12 - 20 GB/s);
                // here every host thread copies its share, and the DMA of one half overlaps the filling of the other
                const int slot = piece_no & 1;
                if (ring_busy[slot]) HIP_TRY(hipEventSynchronize(e->ev_ring[slot]));
                uint8_t *dstp = e->bam_ring[slot].p;
                const uint8_t *srcp = run_src + byte0;
                const int64_t len = byte1 - byte0, blk = (int64_t)1 << 20;
                parallel_chunks((len + blk - 1) / blk, ring_threads, [&](int, int64_t b, int64_t en) {
                    const int64_t lo = b * blk, hi = std::min(len, en * blk);
                    if (hi > lo) std::memcpy(dstp + lo, srcp + lo, (size_t)(hi - lo));
                });
                HIP_TRY(hipMemcpyAsync(d_image.p + byte0, dstp, (size_t)len, hipMemcpyHostToDevice, up));
                HIP_TRY(hipEventRecord(e->ev_ring[slot], up));
                ring_busy[slot] = true;
            } else
                HIP_TRY(hipMemcpyAsync(d_image.p + byte0, run_src + byte0, (size_t)(byte1 - byte0), hipMemcpyHostToDevice, up));
            if (up != st) {
                hipEvent_t x;
                HIP_TRY(hipEventCreateWithFlags(&x, hipEventDisableTiming));
                landed.push_back(x);
                HIP_TRY(hipEventRecord(x, up));
            }
            hipStream_t ks = (piece_no % (naux + 1)) ? e->aux_stream[piece_no % (naux + 1) - 1] : st;
            if (up != st) HIP_TRY(hipStreamWaitEvent(ks, landed.back(), 0));
            if (serial_symbols) hipLaunchKernelGGL(k_bgzf_inflate<false>, dim3((unsigned)(m1 - m0)), dim3(kInflWG), 0, ks, d_image.p, d_members.p, m0, m1, d_stream.p, d_status.p);
            else hipLaunchKernelGGL(k_bgzf_inflate<true>, dim3((unsigned)(m1 - m0)), dim3(kInflWG), 0, ks, d_image.p, d_members.p, m0, m1, d_stream.p, d_status.p);
            // (the piece's CRC check right behind it, on the same stream: it runs while other pieces are still inflated)
            hipLaunchKernelGGL(k_bgzf_crc, dim3((unsigned)(m1 - m0)), dim3(64), 0, ks, d_stream.p, d_members.p, m0, m1, d_crc.p, d_crc.p + 256, d_status.p);
            byte0 = byte1;
            m0 = m1;
        }
        }
        clk.note("every piece copied into the page-locked ring and queued");
        for (int k = 0; k < naux; ++k) {   // the main stream goes on behind all of them
            hipEvent_t x;
            HIP_TRY(hipEventCreateWithFlags(&x, hipEventDisableTiming));
            landed.push_back(x);
            HIP_TRY(hipEventRecord(x, e->aux_stream[k]));
            HIP_TRY(hipStreamWaitEvent(st, x, 0));
        }
        if (uploaded) (*uploaded)(up);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(status.data(), d_status.p, (size_t)nm * 4, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipEventRecord(ev[2], st));
    HIP_TRY(hipStreamSynchronize(st));
    drain.armed = false;   // (the main stream went on behind the side and auxiliary ones: all of them have drained)
    clk.lap("upload + inflate + crc (sync)");
    for (int m = 0; m < nm; ++m)
        if (status[(size_t)m]) {
            if (getenv("PC_BAM_DEBUG")) fprintf(stderr, "[bam] member %d of %d (%u compressed -> %u bytes at %llu): inflate status %u\n", m, nm,
                                                members[(size_t)m].clen, members[(size_t)m].ulen, (unsigned long long)members[(size_t)m].uoff, status[(size_t)m]);
            return fail(PC_ERR_ARG, "%s%s", status[(size_t)m] == (uint32_t)kInfCrc ? "BGZF CRC mismatch in " : "BGZF inflate failed in ", path.c_str());
        }
    d_image.release();
    clk.lap("status check + image release");
    // ---- BAM header (host, from the head of the inflated stream)
    uint64_t first_record = 0;
    uint32_t n_ref = 0;
    // (region read: the header is looked for in the leading members only -- what follows them is the span, from some
    // record in the middle of the file on; a header that does not fit them makes the caller come back with more)
    const size_t header_limit = (span && span_first_member >= 0) ? (size_t)(members[(size_t)span_first_member].uoff + members[(size_t)span_first_member].ulen)
                                                                 : (size_t)total_u;
    {
        std::vector<uint8_t> head;
        size_t want = std::min<size_t>(header_limit, (size_t)1 << 16);
        for (;;) {
            head.resize(want);
            if (want) HIP_TRY(hipMemcpy(head.data(), d_stream.p, want, hipMemcpyDeviceToHost));
            const uint8_t *p = head.data(), *end = p + want;
            bool more = false;
            auto need = [&](size_t k) { if ((size_t)(end - p) < k) { more = true; return false; } return true; };
            bool ok = true;
            if (!need(12)) ok = false;
            if (ok && std::memcmp(p, "BAM\1", 4) != 0) return fail(PC_ERR_ARG, "not a BAM file (bad magic)");
            uint32_t l_text = 0;
            if (ok) { l_text = brd32(p + 4); p += 8; if (!need((size_t)l_text + 4)) ok = false; }
            if (ok) { p += l_text; n_ref = brd32(p); p += 4; }
            b->ref_names.clear(); b->ref_lengths.clear();
            for (uint32_t r = 0; ok && r < n_ref; ++r) {
                if (!need(4)) { ok = false; break; }
                const uint32_t l_name = brd32(p);
                p += 4;
                if (!need((size_t)l_name + 4)) { ok = false; break; }
                b->ref_names.emplace_back((const char *)p, l_name ? l_name - 1 : 0);
                p += l_name;
                b->ref_lengths.push_back((int32_t)brd32(p));
                p += 4;
            }
            if (ok) { first_record = (uint64_t)(p - head.data()); break; }
            if (more && want >= header_limit && span && header_limit < (size_t)total_u + 1 && span->header_bytes < size) return PC_RETRY_HEADER;
            if (!more || want >= header_limit)
                return fail(PC_ERR_ARG, want < 12 ? "not a BAM file (bad magic)" : (b->ref_names.empty() && n_ref == 0 ? "truncated BAM header" : "truncated BAM reference list"));
            want = std::min<size_t>(header_limit, want * 4);
        }
    }
    // ---- record starts: every member guesses its first record start and walks the chain of length prefixes; the
    // host confirms that the walks chain, and restarts the members whose guess did not
    DevBuf<MemberChain> d_chain;
    DevBuf<uint32_t> d_rec_off;
    DevBuf<uint64_t> d_forced;
    rc = d_chain.reserve((size_t)std::max(nm, 1));
    if (rc == PC_OK) rc = d_rec_off.reserve((size_t)std::max(nm, 1) * kMaxRecPerMember);
    if (rc == PC_OK) rc = d_forced.reserve((size_t)std::max(nm, 1));
    if (rc != PC_OK) return rc;
    // region read: the records start where the index says (a record start inside the span's first member) and end at its
    // last chunk's end; a header-only read (no chunk at all) has no records
    uint64_t stop_at = total_u;
    if (span) {
        if (span_first_member < 0) { first_record = total_u; stop_at = total_u; }
        else {
            const uint64_t fr = members[(size_t)span_first_member].uoff + (span->voff_begin & 0xffffu);
            if (fr < first_record || fr > total_u)
                return fail(PC_ERR_ARG, "the index does not belong to this BAM file (a chunk starts inside the header or beyond its member): %s", path.c_str());
            first_record = fr;
            if (span_last_member >= 0) stop_at = members[(size_t)span_last_member].uoff + (span->voff_end & 0xffffu);
            if (stop_at > total_u || stop_at < first_record)
                return fail(PC_ERR_ARG, "the index does not belong to this BAM file (a chunk ends beyond its member): %s", path.c_str());
        }
    }
    std::vector<MemberChain> chain((size_t)nm);
    std::vector<uint64_t> forced((size_t)nm, ~0ull), rec_base((size_t)nm + 1, 0);
    std::vector<uint32_t> nrec_of((size_t)nm, 0u);
    bool truncated = false;
    int64_t nrec = 0;
    if (nm) {
        HIP_TRY(hipMemsetAsync(d_forced.p, 0xff, (size_t)nm * 8, st));
        int from = 0;
        uint64_t expected = first_record;
        for (int round = 0;; ++round) {
            hipLaunchKernelGGL(k_bam_chain, dim3((unsigned)(nm - from)), dim3(64), 0, st, d_stream.p, total_u, d_members.p, nm, from, n_ref, first_record,
                               d_forced.p, d_chain.p, d_rec_off.p, stop_at);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(chain.data() + from, d_chain.p + from, (size_t)(nm - from) * sizeof(MemberChain), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            int redo = -1;
            for (int m = from; m < nm; ++m) {
                const uint64_t begin = members[(size_t)m].uoff, end = std::min<uint64_t>(begin + members[(size_t)m].ulen, stop_at);
                nrec_of[(size_t)m] = 0;
                if (expected >= end) continue;                      // no record starts in this member (or it lies behind the span)
                const MemberChain &mc = chain[(size_t)m];
                if (mc.first != expected) {                         // the guess was off (or there was none): walk again from the right place
                    forced[(size_t)m] = expected;
                    redo = m;
                    break;
                }
                nrec_of[(size_t)m] = mc.nrec;
                if (mc.flags & 2u) { truncated = true; from = nm; break; }   // a length prefix that cannot be: the walk ends here
                expected = mc.next;
            }
            if (redo < 0) break;
            b->chain_restarts += 1;
            HIP_TRY(hipMemcpyAsync(d_forced.p + redo, &forced[(size_t)redo], 8, hipMemcpyHostToDevice, st));
            from = redo;
            if (round > nm + 8) return fail(PC_ERR_STATE, "pc_bam_open: the record chain of %s did not settle", path.c_str());
        }
        if (expected != stop_at) {   // the last record runs past (or stops short of) the end of the stream
            if (span) return fail(PC_ERR_ARG, "the index does not belong to this BAM file (a chunk ends inside a record): %s", path.c_str());
            truncated = true;
        }
        for (int m = 0; m < nm; ++m) rec_base[(size_t)m + 1] = rec_base[(size_t)m] + nrec_of[(size_t)m];
        nrec = (int64_t)rec_base[(size_t)nm];
    } else if (total_u != first_record && !span) truncated = true;
    HIP_TRY(hipEventRecord(ev[3], st));
    clk.lap("header + record chain");
    b->total = nrec;
    if (nrec >= (int64_t)0x7fffffff) return fail(PC_ERR_ARG, "pc_bam_open: more than 2^31-2 records per file are not supported");
    // ---- fields, order checks, columns
    DevBuf<uint64_t> d_rec_base;
    DevBuf<uint32_t> d_rec_member, d_placed, d_runs, d_staged_at, d_run_at, d_wide;
    DevBuf<RecOut> d_recs;
    DevBuf<unsigned long long> d_misc;   // [0] first error (index << 8 | code), [1] mapped, [2] unplaced
    rc = d_misc.reserve(4);
    if (rc != PC_OK) return rc;
    const unsigned long long misc0[4] = {~0ull, 0ull, 0ull, 0ull};
    HIP_TRY(hipMemcpyAsync(d_misc.p, misc0, sizeof(misc0), hipMemcpyHostToDevice, st));
    int64_t n_staged = 0, n_runs = 0;
    if (nrec > 0) {
        std::vector<uint32_t> rec_member((size_t)((nrec + 255) >> 8));
        {
            int m = 0;
            for (size_t g = 0; g < rec_member.size(); ++g) {
                const uint64_t i = (uint64_t)g << 8;
                while (m + 1 < nm && rec_base[(size_t)m + 1] <= i) ++m;
                rec_member[g] = (uint32_t)m;
            }
        }
        rc = d_rec_base.upload(rec_base, st);
        if (rc == PC_OK) rc = d_rec_member.upload(rec_member, st);
        if (rc == PC_OK) rc = d_recs.reserve((size_t)nrec);
        if (rc == PC_OK) rc = d_placed.reserve((size_t)nrec + 1);
        if (rc == PC_OK) rc = d_runs.reserve((size_t)nrec + 1);
        if (rc == PC_OK) rc = d_staged_at.reserve((size_t)nrec + 1);
        if (rc == PC_OK) rc = d_run_at.reserve((size_t)nrec + 1);
        if (rc != PC_OK) return rc;
        const unsigned g256 = (unsigned)((nrec + 255) / 256);
        hipLaunchKernelGGL(k_bam_fields, dim3(g256), dim3(256), 0, st, d_stream.p, total_u, d_members.p, d_rec_base.p, d_chain.p, d_rec_off.p, nm, nrec,
                           n_ref, d_rec_member.p, d_recs.p);
        hipLaunchKernelGGL(k_bam_order, dim3(g256), dim3(256), 0, st, d_recs.p, nrec, d_placed.p, d_misc.p);
        DevBuf<int32_t> d_rtid;
        DevBuf<int64_t> d_rbe;
        if (span) {   // keep what overlaps a requested region (htslib's overlap rule); everything else is as if it were not in the file
            const size_t nr = (size_t)std::max(span->nreg, 0);
            rc = d_rtid.upload(span->tid, nr, st);
            if (rc == PC_OK) rc = d_rbe.reserve(2 * std::max<size_t>(nr, 1));
            if (rc != PC_OK) return rc;
            if (nr) {
                HIP_TRY(hipMemcpyAsync(d_rbe.p, span->beg, nr * 8, hipMemcpyHostToDevice, st));
                HIP_TRY(hipMemcpyAsync(d_rbe.p + nr, span->end, nr * 8, hipMemcpyHostToDevice, st));
            }
            hipLaunchKernelGGL(k_bam_region_filter, dim3(g256), dim3(256), 0, st, d_recs.p, nrec, (int)nr, d_rtid.p, d_rbe.p, d_rbe.p + nr);
        }
        HIP_TRY(hipMemsetAsync(d_placed.p + nrec, 0, 4, st));
        HIP_TRY(hipMemsetAsync(d_runs.p + nrec, 0, 4, st));
        hipLaunchKernelGGL(k_bam_scan_inputs, dim3((unsigned)((nrec + 256 * kScanInputsPerThread - 1) / (256 * kScanInputsPerThread))), dim3(256), 0, st, d_recs.p, nrec, d_placed.p, d_runs.p, d_misc.p + 1);
        {
            size_t tmp_bytes = 0;
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_placed.p, d_staged_at.p, (int)(nrec + 1), st));
            DevBuf<uint8_t> d_tmp;
            rc = d_tmp.reserve(std::max<size_t>(tmp_bytes, 16));
            if (rc != PC_OK) return rc;
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tmp_bytes, d_placed.p, d_staged_at.p, (int)(nrec + 1), st));
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tmp_bytes, d_runs.p, d_run_at.p, (int)(nrec + 1), st));
            uint32_t tot[2] = {0, 0};
            unsigned long long misc[3] = {0, 0, 0};
            HIP_TRY(hipMemcpyAsync(&tot[0], d_staged_at.p + nrec, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(&tot[1], d_run_at.p + nrec, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(misc, d_misc.p, sizeof(misc), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));   // (d_tmp goes out of scope)
            n_staged = tot[0]; n_runs = tot[1];
            b->mapped = (int64_t)misc[1]; b->unplaced = (int64_t)misc[2];
            if (misc[0] != ~0ull) {
                switch ((int)(misc[0] & 0xffu)) {
                case kRecTidRange: return fail(PC_ERR_ARG, "BAM record with reference id out of range");
                case kRecNegPos: return fail(PC_ERR_ARG, "placed BAM record with a negative position");
                case kRecUnsorted: return fail(PC_ERR_UNSORTED, "BAM file is not coordinate sorted: %s", path.c_str());
                case kRecCigarOverrun: return fail(PC_ERR_ARG, "corrupt BAM record (cigar overruns block)");
                case kRecUnknownOp: return fail(PC_ERR_ARG, "unknown CIGAR operation in %s", path.c_str());
                case kRecEndBeyond: return fail(PC_ERR_ARG, "alignment ends beyond 2^31 - 1");
                case kRecTooLong: return fail(PC_ERR_ARG, "alignment with more than 2^31 - 1 aligned positions");
                case kRecDeletionOrder: return fail(PC_ERR_ARG, "alignment starting with a deletion breaks coordinate order; not supported");
                default: truncated = true; break;   // kRecTruncated / kRecBadSize: reported below, after every other defect
                }
            }
        }
        if (truncated) return fail(PC_ERR_ARG, "truncated BAM record");
        rc = b->tid.reserve((size_t)std::max<int64_t>(n_staged, 1));
        if (rc == PC_OK) rc = b->pos.reserve((size_t)std::max<int64_t>(n_staged, 1));
        if (rc == PC_OK) rc = b->alen.reserve((size_t)std::max<int64_t>(n_staged, 1));
        if (rc == PC_OK) rc = b->flags.reserve((size_t)std::max<int64_t>(n_staged, 1));
        if (rc == PC_OK) rc = b->nblk.reserve((size_t)std::max<int64_t>(n_staged, 1));
        if (rc == PC_OK) rc = b->flag16.reserve((size_t)std::max<int64_t>(n_staged, 1));
        if (rc == PC_OK) rc = b->mapq.reserve((size_t)std::max<int64_t>(n_staged, 1));
        if (rc == PC_OK) rc = b->lseq.reserve((size_t)std::max<int64_t>(n_staged, 1));
        if (rc == PC_OK) rc = b->nh.reserve((size_t)std::max<int64_t>(n_staged, 1));
        if (rc == PC_OK) rc = b->blk_start.reserve((size_t)std::max<int64_t>(n_runs, 1));
        if (rc == PC_OK) rc = b->blk_len.reserve((size_t)std::max<int64_t>(n_runs, 1));
        if (rc == PC_OK) rc = d_wide.reserve((size_t)std::max<int64_t>(n_staged, 1));
        if (rc != PC_OK) return rc;
        HIP_TRY(hipMemsetAsync(d_wide.p, 0, (size_t)std::max<int64_t>(n_staged, 1) * 4, st));
        hipLaunchKernelGGL(k_bam_columns, dim3(g256), dim3(256), 0, st, d_stream.p, d_members.p, d_rec_base.p, d_rec_off.p, nm, d_rec_member.p, d_recs.p,
                           nrec, d_staged_at.p, d_run_at.p, b->tid.p, b->pos.p, b->alen.p, b->flags.p, b->nblk.p, b->blk_start.p, b->blk_len.p, d_wide.p,
                           b->flag16.p, b->mapq.p, b->lseq.p, b->nh.p);
        HIP_TRY(hipGetLastError());
        // wide records (beyond the 16-bit / 8-bit columns): rare -- their staged indices are found from the markers on
        // the host side of pc_bam_read; the true values are read back here, record by record
        {
            // count the flagged records (a sum reduction through the scan buffers would do; a plain read-back of the
            // flags is only paid when the file has any: probe with a device-side total first)
            DevBuf<uint32_t> d_wsum;
            rc = d_wsum.reserve((size_t)n_staged + 1);
            if (rc != PC_OK) return rc;
            size_t tmp_bytes = 0;
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_wide.p, d_wsum.p, (int)std::max<int64_t>(n_staged, 1), st));
            DevBuf<uint8_t> d_tmp;
            rc = d_tmp.reserve(std::max<size_t>(tmp_bytes, 16));
            if (rc != PC_OK) return rc;
            uint32_t last_sum = 0, last_flag = 0;
            if (n_staged > 0) {
                HIP_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tmp_bytes, d_wide.p, d_wsum.p, (int)n_staged, st));
                HIP_TRY(hipMemcpyAsync(&last_sum, d_wsum.p + (n_staged - 1), 4, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipMemcpyAsync(&last_flag, d_wide.p + (n_staged - 1), 4, hipMemcpyDeviceToHost, st));
            }
            HIP_TRY(hipStreamSynchronize(st));
            const uint32_t nwide = last_sum + last_flag;
            if (nwide) {
                std::vector<uint32_t> wf((size_t)n_staged), sa((size_t)nrec);
                std::vector<RecOut> recs((size_t)nrec);
                HIP_TRY(hipMemcpy(wf.data(), d_wide.p, (size_t)n_staged * 4, hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(sa.data(), d_staged_at.p, (size_t)nrec * 4, hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(recs.data(), d_recs.p, (size_t)nrec * sizeof(RecOut), hipMemcpyDeviceToHost));
                for (int64_t i = 0; i < nrec; ++i)
                    if (recs[(size_t)i].placed == 1 && wf[sa[(size_t)i]]) {
                        b->wide_idx.push_back((int64_t)sa[(size_t)i]);
                        b->wide_alen.push_back((int32_t)recs[(size_t)i].L);
                        b->wide_nblk.push_back((int32_t)recs[(size_t)i].nruns);
                    }
            }
        }
    } else if (truncated) return fail(PC_ERR_ARG, "truncated BAM record");
    HIP_TRY(hipEventRecord(ev[4], st));
    HIP_TRY(hipStreamSynchronize(st));
    clk.lap("fields + scans + columns");
    b->n = n_staged; b->nrun = n_runs;
    for (int k = 0; k < 4; ++k) b->ms[k] = ms_between(ev[k], ev[k + 1]);
    guard.b = nullptr;
    *out = b;
    return PC_OK;
}

int pc_bam_counts(pc_bam *b, int64_t *counts) {
    if (!b || !counts) return fail(PC_ERR_ARG, "pc_bam_counts: bad arguments");
    counts[0] = b->n; counts[1] = b->nrun; counts[2] = b->mapped; counts[3] = b->total; counts[4] = (int64_t)b->wide_idx.size();
    counts[5] = b->members; counts[6] = b->inflated_bytes; counts[7] = b->chain_restarts;
    return PC_OK;
}
int pc_bam_timing(pc_bam *b, double *ms4) {
    if (!b || !ms4) return fail(PC_ERR_ARG, "pc_bam_timing: bad arguments");
    for (int k = 0; k < 4; ++k) ms4[k] = b->ms[k];
    return PC_OK;
}
int pc_bam_nref(pc_bam *b) { return b ? (int)b->ref_names.size() : -1; }
const char *pc_bam_ref_name(pc_bam *b, int i) { return (b && i >= 0 && i < (int)b->ref_names.size()) ? b->ref_names[(size_t)i].c_str() : nullptr; }
int32_t pc_bam_ref_length(pc_bam *b, int i) { return (b && i >= 0 && i < (int)b->ref_lengths.size()) ? b->ref_lengths[(size_t)i] : -1; }

int pc_bam_read(pc_bam *b, int32_t *tid, int32_t *pos, uint16_t *alen, uint8_t *flags, uint8_t *nblk, int32_t *blk_start, int32_t *blk_len,
                int64_t *wide_idx, int32_t *wide_alen, int32_t *wide_nblk) {
    if (!b) return fail(PC_ERR_ARG, "pc_bam_read: NULL handle");
    if (b->n > 0 && (!tid || !pos || !alen || !flags || !nblk)) return fail(PC_ERR_ARG, "pc_bam_read: NULL array");
    if (b->nrun > 0 && (!blk_start || !blk_len)) return fail(PC_ERR_ARG, "pc_bam_read: NULL run array");
    if (!b->wide_idx.empty() && (!wide_idx || !wide_alen || !wide_nblk)) return fail(PC_ERR_ARG, "pc_bam_read: NULL wide array");
    HIP_TRY(hipSetDevice(b->e->device));
    hipStream_t st = b->e->stream;
    BamClock rclk;
    const size_t n = (size_t)b->n, m = (size_t)b->nrun;
    // (through the ring of page-locked pieces when the columns are large: the caller's arrays are pageable, see TransferRing)
    HIP_TRY(hipStreamSynchronize(st));
    std::vector<TransferJob> jobs;
    if (n) {
        jobs.push_back({tid, b->tid.p, n * 4});
        jobs.push_back({pos, b->pos.p, n * 4});
        jobs.push_back({alen, b->alen.p, n * 2});
        jobs.push_back({flags, b->flags.p, n});
        jobs.push_back({nblk, b->nblk.p, n});
    }
    if (m) {
        jobs.push_back({blk_start, b->blk_start.p, m * 4});
        jobs.push_back({blk_len, b->blk_len.p, m * 4});
    }
    {
        const int rc = TransferRing::of(b->e->device).run(b->e->device, jobs, TransferRing::kPiece, false, true);
        if (rc != PC_OK) return rc;
    }
    rclk.lap("columns to the host");
    for (size_t k = 0; k < b->wide_idx.size(); ++k) { wide_idx[k] = b->wide_idx[k]; wide_alen[k] = b->wide_alen[k]; wide_nblk[k] = b->wide_nblk[k]; }
    return PC_OK;
}

int pc_bam_read_sam(pc_bam *b, uint16_t *flag, uint8_t *mapq, int32_t *lseq) {
    if (!b) return fail(PC_ERR_ARG, "pc_bam_read_sam: NULL handle");
    HIP_TRY(hipSetDevice(b->e->device));
    hipStream_t st = b->e->stream;
    const size_t n = (size_t)b->n;
    HIP_TRY(hipStreamSynchronize(st));
    std::vector<TransferJob> jobs;
    if (n && flag) jobs.push_back({flag, b->flag16.p, n * 2});
    if (n && mapq) jobs.push_back({mapq, b->mapq.p, n});
    if (n && lseq) jobs.push_back({lseq, b->lseq.p, n * 4});
    return TransferRing::of(b->e->device).run(b->e->device, jobs, TransferRing::kPiece, false, true);
}

int pc_bam_read_nh(pc_bam *b, uint16_t *nh) {
    if (!b) return fail(PC_ERR_ARG, "pc_bam_read_nh: NULL handle");
    HIP_TRY(hipSetDevice(b->e->device));
    HIP_TRY(hipStreamSynchronize(b->e->stream));
    std::vector<TransferJob> jobs;
    if (b->n && nh) jobs.push_back({nh, b->nh.p, (size_t)b->n * 2});
    return TransferRing::of(b->e->device).run(b->e->device, jobs, TransferRing::kPiece, false, true);
}

static int add_alignment_bam_impl(pc_engine *e, const void *image, int64_t size, const char *name, int64_t *mapped, const UploadedHook *uploaded,
                                  const BamSpan *span = nullptr);

int pc_add_alignment_bam(pc_engine *e, const void *image, int64_t size, const char *name, int64_t *mapped) {
    return add_alignment_bam_impl(e, image, size, name, mapped, nullptr);
}

// a region read comes back for a larger slice of the file's head when the header did not fit the first one
static int bam_open_span_retry(pc_engine *e, const void *image, int64_t size, const char *name, pc_bam **out, const UploadedHook *uploaded, const BamSpan *span) {
    if (!span) return bam_open_impl(e, image, size, name, out, uploaded, nullptr);
    BamSpan sp = *span;
    for (;;) {
        const int rc = bam_open_impl(e, image, size, name, out, uploaded, &sp);
        if (rc != PC_RETRY_HEADER) return rc;
        sp.header_bytes = std::min<int64_t>(size, sp.header_bytes * 16);
    }
}

static int add_alignment_bam_impl(pc_engine *e, const void *image, int64_t size, const char *name, int64_t *mapped, const UploadedHook *uploaded,
                                  const BamSpan *span) {
    pc_bam *b = nullptr;
    int rc = bam_open_span_retry(e, image, size, name, &b, uploaded, span);
    if (rc != PC_OK) return rc;
    struct Closer { pc_bam *b; ~Closer() { pc_bam_close(b); } } closer{b};
    PoolScope pool_scope(&e->pool);
    const int64_t n = b->n, m = b->nrun, nw = (int64_t)b->wide_idx.size();
    const int ntid = std::max(1, (int)b->ref_names.size());
    if (mapped) *mapped = b->mapped;
    // the decoder's columns never leave HBM
    // (PC_BAM_STAGE_HOST=1: read them back and hand them over as host arrays, as a caller of pc_bam_read + pc_add_alignment_file does)
    if (!getenv("PC_BAM_STAGE_HOST")) {
        DevBuf<uint32_t> d_wr;
        DevBuf<uint2> d_wv;
        if (nw) {
            std::vector<uint32_t> wr((size_t)nw);
            std::vector<uint2> wv((size_t)nw);
            for (int64_t k = 0; k < nw; ++k) { wr[(size_t)k] = (uint32_t)b->wide_idx[(size_t)k]; wv[(size_t)k] = make_uint2((uint32_t)b->wide_alen[(size_t)k], (uint32_t)b->wide_nblk[(size_t)k]); }
            rc = d_wr.upload(wr, e->stream);
            if (rc == PC_OK) rc = d_wv.upload(wv, e->stream);
            if (rc != PC_OK) return rc;
            HIP_TRY(hipStreamSynchronize(e->stream));   // (the host vectors go out of scope)
        }
        pcstage::DevCols dc;
        dc.tid = b->tid.p; dc.pos = b->pos.p; dc.alen = b->alen.p; dc.flags = b->flags.p; dc.nblk = b->nblk.p;
        dc.blk_start = b->blk_start.p; dc.blk_len = b->blk_len.p;
        dc.wide_rec = d_wr.p; dc.wide_val = d_wv.p; dc.n_wide = nw;
        rc = stage_file(e, n, ntid, nullptr, nullptr, nullptr, nullptr, nullptr, m, nullptr, nullptr, nw, b->wide_idx.data(), b->wide_alen.data(),
                        b->wide_nblk.data(), &dc);
        if (rc != PC_OK) return rc;
        // the FLAG / MAPQ columns stay with the staged file (no copy: the decoder's blocks change hands)
        StagedFile *sf = e->files.back();
        sf->sam_flag.swap(b->flag16);
        sf->sam_mapq.swap(b->mapq);
        sf->have_sam = true;
        sf->sam_nh.swap(b->nh);
        sf->have_nh = true;
        return e->filter_on() ? apply_flag_filter(e, sf) : PC_OK;
    }
    std::vector<int32_t> tid((size_t)n), pos((size_t)n), bs((size_t)m), bl((size_t)m), wa((size_t)nw), wn((size_t)nw);
    std::vector<uint16_t> alen((size_t)n);
    std::vector<uint8_t> flags((size_t)n), nblk((size_t)n);
    std::vector<int64_t> wi((size_t)nw);
    rc = pc_bam_read(b, tid.data(), pos.data(), alen.data(), flags.data(), nblk.data(), bs.data(), bl.data(), wi.data(), wa.data(), wn.data());
    if (rc != PC_OK) return rc;
    std::vector<uint16_t> f16((size_t)n);
    std::vector<uint8_t> mq((size_t)n);
    rc = pc_bam_read_sam(b, f16.data(), mq.data(), nullptr);
    if (rc != PC_OK) return rc;
    rc = pc_add_alignment_file_wide(e, n, ntid, tid.data(), pos.data(), alen.data(), flags.data(), nblk.data(), m, bs.data(), bl.data(),
                                    nw, wi.data(), wa.data(), wn.data());
    if (rc != PC_OK) return rc;
    rc = pc_set_alignment_sam(e, (int)e->files.size() - 1, n, f16.data(), mq.data());
    if (rc != PC_OK) return rc;
    std::vector<uint16_t> nhv((size_t)n);
    rc = pc_bam_read_nh(b, nhv.data());
    if (rc != PC_OK) return rc;
    return pc_set_alignment_nh(e, (int)e->files.size() - 1, n, nhv.data());
}

namespace {
// A file mapped for one call: every host thread touches its share of the pages (soft faults in parallel: 578 MB in ~2 ms
// instead of the 15 of MAP_POPULATE's one thread), and the mapping is taken down by a helper thread as soon as the
// image has been uploaded -- while the caller's thread waits for the GPU.
struct MappedFile {
    void *p = nullptr;
    size_t size = 0;
    // (a region read maps only, and faults what it uploads)
    // touch_mode 1: fault every page in up front (all host threads); 0: map only; -1 (whole-file reads): up front for a
    // file that is uploaded straight from the mapping (below two upload pieces: the runtime's one staging thread would take
    // the faults one by one), map only for a larger one -- its pieces are copied into the page-locked ring by all host
    // threads, which take the faults as they go, beside the GPU's work (the 2.9 GB file of 10^8 aligner-like records:
    // file -> staged 169 - 180 ms with the pages touched up front, 147 - 149 without; PC_BAM_TOUCH=0 / 1 forces either)
    int open(const char *path, int touch_mode = -1) {
        bool touch = touch_mode > 0;
        const char *env = getenv("PC_BAM_TOUCH");
        const int forced = (touch_mode < 0 && env) ? (atoi(env) != 0 ? 1 : 0) : -1;
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return fail(PC_ERR_ARG, "cannot open %s: %s", path, strerror(errno));
        struct stat sb;
        if (fstat(fd, &sb) != 0) { ::close(fd); return fail(PC_ERR_ARG, "cannot stat %s: %s", path, strerror(errno)); }
        size = (size_t)sb.st_size;
        if (touch_mode < 0) touch = forced >= 0 ? forced != 0 : size < ((size_t)128 << 20);
        if (size) {
            p = mmap(nullptr, size, PROT_READ, MAP_SHARED, fd, 0);
            if (p == MAP_FAILED) { p = nullptr; ::close(fd); return fail(PC_ERR_NOMEM, "cannot map %s: %s", path, strerror(errno)); }
            if (touch) {
                (void)madvise(p, size, MADV_WILLNEED);
                const int64_t pages = (int64_t)((size + 4095) / 4096);
                const volatile uint8_t *q = (const volatile uint8_t *)p;
                parallel_chunks(pages, std::min(usable_cpus(), 32), [&](int, int64_t b, int64_t en) {
                    uint8_t acc = 0;
                    for (int64_t k = b; k < en; ++k) acc ^= q[(size_t)k * 4096];
                    (void)acc;
                });
            }
        }
        ::close(fd);
        return PC_OK;
    }
    // the mapping goes as soon as the image has crossed PCIe -- on a helper thread, while the caller's thread waits for
    // the GPU (munmap holds the address-space lock of the process: done later, it would stall the caller's next page faults)
    std::thread helper;
    void release_behind(hipStream_t up, int device) {
        if (!p || helper.joinable()) return;
        void *q = p;
        const size_t n = size;
        p = nullptr;
        try {
            helper = std::thread([q, n, up, device]() {
                if (hipSetDevice(device) == hipSuccess) (void)hipStreamSynchronize(up);
                (void)munmap(q, n);
            });
        } catch (...) { p = q; }
    }
    ~MappedFile() {
        if (helper.joinable()) helper.join();
        if (p) (void)munmap(p, size);
    }
};
} // namespace

int pc_bam_open_path(pc_engine *e, const char *path, pc_bam **out) {
    if (!e || !path || !out) return fail(PC_ERR_ARG, "pc_bam_open_path: bad arguments");
    MappedFile mf;
    const int rc = mf.open(path);
    if (rc != PC_OK) return rc;
    const int device = e->device;
    const UploadedHook hook = [&mf, device](hipStream_t up) { mf.release_behind(up, device); };
    return bam_open_impl(e, mf.p, (int64_t)mf.size, path, out, &hook);
}

int pc_add_alignment_bam_path(pc_engine *e, const char *path, int64_t *mapped) {
    if (!e || !path) return fail(PC_ERR_ARG, "pc_add_alignment_bam_path: bad arguments");
    MappedFile mf;
    const int rc = mf.open(path);
    if (rc != PC_OK) return rc;
    const int device = e->device;
    const UploadedHook hook = [&mf, device](hipStream_t up) { mf.release_behind(up, device); };
    return add_alignment_bam_impl(e, mf.p, (int64_t)mf.size, path, mapped, &hook);
}

static int span_args(const char *what, uint64_t voff_begin, uint64_t voff_end, int nreg, const int32_t *tid, const int64_t *beg, const int64_t *end, BamSpan &sp) {
    if (voff_end < voff_begin || nreg < 0 || (nreg > 0 && (!tid || !beg || !end))) return fail(PC_ERR_ARG, "%s: bad span / regions", what);
    for (int k = 0; k < nreg; ++k) {
        if (tid[k] < 0 || end[k] <= beg[k]) return fail(PC_ERR_ARG, "%s: region %d is empty or has no reference id", what, k);
        if (k > 0 && (tid[k] < tid[k - 1] || (tid[k] == tid[k - 1] && beg[k] < end[k - 1])))
            return fail(PC_ERR_ARG, "%s: regions must be ascending by (reference id, start) and must not overlap", what);
    }
    sp.voff_begin = voff_begin; sp.voff_end = voff_end; sp.nreg = nreg; sp.tid = tid; sp.beg = beg; sp.end = end;
    if (const char *env = getenv("PC_BAM_HEADER_BYTES")) sp.header_bytes = std::max<int64_t>(1, atoll(env));   // (tests: a header longer than the first slice)
    return PC_OK;
}

int pc_bam_open_span(pc_engine *e, const char *path, uint64_t voff_begin, uint64_t voff_end, int nreg, const int32_t *tid, const int64_t *beg,
                     const int64_t *end, pc_bam **out) {
    if (!e || !path || !out) return fail(PC_ERR_ARG, "pc_bam_open_span: bad arguments");
    BamSpan sp;
    int rc = span_args("pc_bam_open_span", voff_begin, voff_end, nreg, tid, beg, end, sp);
    if (rc != PC_OK) return rc;
    MappedFile mf;
    rc = mf.open(path, 0);
    if (rc != PC_OK) return rc;
    return bam_open_span_retry(e, mf.p, (int64_t)mf.size, path, out, nullptr, &sp);
}

int pc_add_alignment_bam_span(pc_engine *e, const char *path, uint64_t voff_begin, uint64_t voff_end, int nreg, const int32_t *tid,
                              const int64_t *beg, const int64_t *end, int64_t *mapped) {
    if (!e || !path) return fail(PC_ERR_ARG, "pc_add_alignment_bam_span: bad arguments");
    BamSpan sp;
    int rc = span_args("pc_add_alignment_bam_span", voff_begin, voff_end, nreg, tid, beg, end, sp);
    if (rc != PC_OK) return rc;
    MappedFile mf;
    rc = mf.open(path, 0);
    if (rc != PC_OK) return rc;
    return add_alignment_bam_impl(e, mf.p, (int64_t)mf.size, path, mapped, nullptr, &sp);
}

} // extern "C"
