// upload_probe.hip -- how fast do pageable host arrays reach HBM?  (round 5, staging of caller-owned columns)
//   hipcc --offload-arch=gfx950 -O2 -pthread scripts/ubench/upload_probe.hip -o /tmp/upload_probe && /tmp/upload_probe [GB]
// Ways measured, each on the same pageable buffer (malloc, touched) of the given size:
//   pageable      one hipMemcpy
//   threads(T)    T host threads, each a hipMemcpyAsync of its share on a stream of its own
//   ring(T,S)     T threads copy S-MB pieces into a ring of page-locked buffers; each piece goes up with hipMemcpyAsync
//   register      hipHostRegister + one hipMemcpyAsync + hipHostUnregister (each timed)
//   pinned        the same bytes from page-locked memory (the PCIe rate itself)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void par_memcpy(char *dst, const char *src, size_t n, int T) {
    std::vector<std::thread> th;
    const size_t q = (n + T - 1) / T;
    for (int t = 0; t < T; ++t) {
        const size_t b = std::min(n, (size_t)t * q), e = std::min(n, b + q);
        th.emplace_back([=] { if (e > b) memcpy(dst + b, src + b, e - b); });
    }
    for (auto &x : th) x.join();
}

static void par_memset(char *dst, size_t n, int T) {
    std::vector<std::thread> th;
    const size_t q = (n + T - 1) / T;
    for (int t = 0; t < T; ++t) {
        const size_t b = std::min(n, (size_t)t * q), e = std::min(n, b + q);
        th.emplace_back([=] { if (e > b) memset(dst + b, t + 1, e - b); });
    }
    for (auto &x : th) x.join();
}

int main(int argc, char **argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 2.0;
    const size_t n = (size_t)(gb * (1u << 30));
    auto fresh = [&](char *old) {   // a buffer the runtime has not seen: every way is measured on first contact
        free(old);
        char *q = (char *)malloc(n);
        par_memset(q, n, 16);
        return q;
    };
    char *h = fresh(nullptr);
    char *d = nullptr;
    CK(hipMalloc((void **)&d, n));
    CK(hipMemset(d, 0, n));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; ++rep) {
        if (rep < 2) h = fresh(h);
        double t0 = now();
        CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice));
        double t1 = now();
        printf("pageable            %7.1f ms  %6.1f GB/s\n", (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
    }
    for (int T : {2, 4, 8, 16}) {
        std::vector<hipStream_t> st((size_t)T);
        for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (int rep = 0; rep < 2; ++rep) {
            h = fresh(h);
            double t0 = now();
            std::vector<std::thread> th;
            const size_t q = (n + T - 1) / T;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    const size_t b = std::min(n, (size_t)t * q), e = std::min(n, b + q);
                    CK(hipMemcpyAsync(d + b, h + b, e - b, hipMemcpyHostToDevice, st[(size_t)t]));
                    CK(hipStreamSynchronize(st[(size_t)t]));
                });
            for (auto &x : th) x.join();
            double t1 = now();
            printf("threads(%2d)         %7.1f ms  %6.1f GB/s\n", T, (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
        }
        for (auto &s : st) CK(hipStreamDestroy(s));
    }
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int T : {4, 8, 16})
        for (size_t mb : {8, 32, 128}) {
            const int R = 4;
            const size_t piece = mb << 20;
            char *ring[R];
            hipEvent_t ev[R];
            for (int k = 0; k < R; ++k) { CK(hipHostMalloc((void **)&ring[k], piece, hipHostMallocDefault)); CK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming)); }
            for (int rep = 0; rep < 2; ++rep) {
                h = fresh(h);
                double t0 = now();
                int k = 0;
                for (size_t off = 0; off < n; off += piece, ++k) {
                    const size_t len = std::min(piece, n - off);
                    if (k >= R) CK(hipEventSynchronize(ev[k % R]));
                    par_memcpy(ring[k % R], h + off, len, T);
                    CK(hipMemcpyAsync(d + off, ring[k % R], len, hipMemcpyHostToDevice, s));
                    CK(hipEventRecord(ev[k % R], s));
                }
                CK(hipStreamSynchronize(s));
                double t1 = now();
                if (rep) printf("ring(T=%2d,%3zu MB)   %7.1f ms  %6.1f GB/s\n", T, mb, (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
            }
            for (int k = 0; k < R; ++k) { CK(hipHostFree(ring[k])); CK(hipEventDestroy(ev[k])); }
        }
    for (int rep = 0; rep < 2; ++rep) {
        h = fresh(h);
        double t0 = now();
        CK(hipHostRegister(h, n, hipHostRegisterDefault));
        double t1 = now();
        CK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        double t2 = now();
        CK(hipHostUnregister(h));
        double t3 = now();
        printf("register            %7.1f ms + copy %7.1f ms (%6.1f GB/s) + unregister %7.1f ms  -> %6.1f GB/s overall\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3,
               n / (t2 - t1) / 1e9, (t3 - t2) * 1e3, n / (t3 - t0) / 1e9);
    }
    {   // register in pieces on worker threads while earlier pieces are in flight
        for (int T : {4, 8}) {
            const size_t piece = (size_t)64 << 20;
            const size_t np = (n + piece - 1) / piece;
            h = fresh(h);
            double t0 = now();
            std::vector<std::thread> th;
            std::vector<int> ready(np, 0);
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    for (size_t p = (size_t)t; p < np; p += (size_t)T) {
                        CK(hipHostRegister(h + p * piece, std::min(piece, n - p * piece), hipHostRegisterDefault));
                        __atomic_store_n(&ready[p], 1, __ATOMIC_RELEASE);
                    }
                });
            for (size_t p = 0; p < np; ++p) {
                while (!__atomic_load_n(&ready[p], __ATOMIC_ACQUIRE)) std::this_thread::yield();
                CK(hipMemcpyAsync(d + p * piece, h + p * piece, std::min(piece, n - p * piece), hipMemcpyHostToDevice, s));
            }
            CK(hipStreamSynchronize(s));
            double t1 = now();
            for (auto &x : th) x.join();
            for (size_t p = 0; p < np; ++p) CK(hipHostUnregister(h + p * piece));
            double t2 = now();
            printf("register pieces(T=%d) %6.1f ms (%6.1f GB/s) + unregister %7.1f ms\n", T, (t1 - t0) * 1e3, n / (t1 - t0) / 1e9, (t2 - t1) * 1e3);
        }
    }
    {
        char *p = nullptr;
        CK(hipHostMalloc((void **)&p, n, hipHostMallocDefault));
        par_memcpy(p, h, n, 8);
        for (int rep = 0; rep < 2; ++rep) {
            double t0 = now();
            CK(hipMemcpyAsync(d, p, n, hipMemcpyHostToDevice, s));
            CK(hipStreamSynchronize(s));
            double t1 = now();
            printf("pinned              %7.1f ms  %6.1f GB/s\n", (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
        }
        double t0 = now();
        par_memcpy(p, h, n, 16);
        double t1 = now();
        printf("host memcpy(16 thr) %7.1f ms  %6.1f GB/s\n", (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
        CK(hipHostFree(p));
    }
    return 0;
}
