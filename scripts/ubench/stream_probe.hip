// Micro-benchmark: how fast can a gfx950 stream 8-byte records with the access pattern of
// k_hist_point (one contiguous chunk per workgroup), and what do LDS atomics add?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int U, int SCRAMBLE = 0>
__global__ __launch_bounds__(256) void probe(const u32x4* __restrict__ rec4, int64_t npairs, int64_t chunk_pairs, int G, unsigned* out) {
    extern __shared__ unsigned bins[];
    for (int i = threadIdx.x; i < 2 * G; i += 256) bins[i] = 0;
    __syncthreads();
    const unsigned nchunks = (unsigned)((npairs + chunk_pairs - 1) / chunk_pairs);
    if (SCRAMBLE && blockIdx.x >= nchunks) return;
    const unsigned cid = SCRAMBLE ? (unsigned)(((unsigned long long)blockIdx.x * 2654435761ull) % nchunks) : blockIdx.x;
    const int64_t lo = (int64_t)cid * chunk_pairs;
    const int64_t hi = lo + chunk_pairs < npairs ? lo + chunk_pairs : npairs;
    unsigned acc = 0;
    const unsigned win = rec4[lo < npairs ? lo : 0].x;
    for (int64_t base = lo; base < hi; base += 256 * U) {
        u32x4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { int64_t j = base + u * 256 + threadIdx.x; r[u] = j < hi ? rec4[j] : (u32x4){0,0,0,0}; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (MODE == 0) { acc ^= r[u].x ^ r[u].y ^ r[u].z ^ r[u].w; }
            else {
                // MODE 1: LDS atomic on (pos - win) & (G-1), strand picks the half ; MODE 2: plain ds_write instead
                unsigned d0 = (r[u].x - win + 12) & (G - 1), d1 = (r[u].z - win + 12) & (G - 1);
                unsigned a0 = ((r[u].y >> 16) & 1) * G + d0, a1 = ((r[u].w >> 16) & 1) * G + d1;
                if (MODE == 1) { atomicAdd(&bins[a0], 1u); atomicAdd(&bins[a1], 1u); }
                else { bins[a0] = d0; bins[a1] = d1; }
            }
        }
    }
    __syncthreads();
    if (MODE == 0) { if (acc == 0x12345678u) out[0] = acc; }
    else { unsigned s = 0; for (int i = threadIdx.x; i < 2 * G; i += 256) s += bins[i]; if (s == 0x12345678u) out[0] = s; }
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 100000000;
    const double density = argc > 2 ? atof(argv[2]) : 8.2; // records per position
    std::vector<uint2> h(n + 2);
    uint64_t s = 88172645463325252ull;
    for (int64_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        h[i].x = (unsigned)(i / density); h[i].y = (25 + (s % 10)) | (((s >> 20) & 1) << 16) | (1u << 24); }
    uint2* d; unsigned* out;
    CK(hipMalloc(&d, (n + 2) * 8)); CK(hipMalloc(&out, 64));
    CK(hipMemcpy(d, h.data(), (n + 2) * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int64_t npairs = (n + 1) / 2;
    auto run = [&](const char* name, auto kern, int64_t chunk_records, int G) {
        int64_t chunk_pairs = chunk_records / 2;
        unsigned grid = (unsigned)((npairs + chunk_pairs - 1) / chunk_pairs);
        size_t lds = 2 * G * 4;
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, (const u32x4*)d, npairs, chunk_pairs, G, out);
        CK(hipEventRecord(e0));
        for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, (const u32x4*)d, npairs, chunk_pairs, G, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        printf("%-28s chunk=%7lld G=%5d grid=%6u  %.3f ms  %.0f GB/s\n", name, (long long)chunk_records, G, grid, ms, n * 8.0 / ms / 1e6);
    };
    for (int64_t chunk : {8192, 16384, 65536}) {
        for (int G : {1024, 2048, 4096}) {
            run("stream only U=4", probe<0, 4>, chunk, G);
            run("stream + ds_add U=4", probe<1, 4>, chunk, G);
            run("stream + ds_write U=4", probe<2, 4>, chunk, G);
        }
    }
    run("scrambled stream only", probe<0, 4, 1>, 16384, 2048);
    run("scrambled stream+ds_add", probe<1, 4, 1>, 16384, 2048);
    run("scrambled stream only 8k", probe<0, 4, 1>, 8192, 2048);
    run("stream only U=8", probe<0, 8>, 16384, 2048);
    run("stream + ds_add U=8", probe<1, 8>, 16384, 2048);
    run("stream only U=2", probe<0, 2>, 16384, 2048);
    run("stream + ds_add U=2", probe<1, 2>, 16384, 2048);
    return 0;
}
