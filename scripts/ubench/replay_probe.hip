// replay_probe.hip -- what bounds the ordered float64 replay of k_center?  (scratch experiment)
//
// Every wave walks a list of N 16-byte entries {a, m, 1/m} and adds 1/m to its lanes' accumulators
// where the lane's position lies in [a, a+m) -- the inner loop of the center kernel -- in several
// forms.  Reported: cycles per wave per list entry at the given occupancy (waves per SIMD), from
// the kernel's wall time.
//   V0  broadcast ds_read_b128 + select (v_cndmask x2) + v_add_f64      (round-1 kernel)
//   V1  broadcast ds_read_b128 + v_cmpx / exec-masked v_add_f64
//   V2  per-lane cursors, uniform odd stride (conflict-free) + select
//   V3  per-lane cursors, random spacing (bank conflicts) + select
//   V4  per-lane uniform stride, 8-byte entry {a,m} + 1/m from an LDS table (two ds_read_b64) + select
//   V5  no LDS: entries made up arithmetically (VALU cost of test + add alone)
//   V6  per-lane uniform stride + v_cmpx form
//   V7  V2 with two positions per lane (one entry read serves two accumulators)
//   V8  broadcast, group-aware: p - a once per 32 entries, per entry compare + ONE select (a miss adds
//       the subnormal {low word of 1/m, 0} instead of 0.0) + v_add_f64
//   V9  as V8 with the full two-word select
//   V10 as V8, entries read as {m} (ds_read_b32) + {1/m} (ds_read_b64) from two arrays
//   V11 no LDS: entries come through the scalar path (s_load from a global list, wave-uniform
//       address), SGPR operands in the vector instructions, select form
//   V12 as V11 with v_cmpx / exec-masked v_add_f64 (3 vector instructions per entry)
// build: hipcc --offload-arch=gfx950 -O3 -o replay_probe replay_probe.hip ; run: ./replay_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((aligned(16))) Entry { int a, m; double val; };
constexpr int kN = 1024;       // list entries in LDS per block
constexpr int kWG = 256;

__device__ __forceinline__ Entry ld16(const Entry *q) {
    const u32x4 v = *(const u32x4 *)q;
    Entry e;
    e.a = (int)v.x; e.m = (int)v.y;
    e.val = __longlong_as_double((long long)(((unsigned long long)v.w << 32) | v.z));
    return e;
}

template <int V>
__global__ __launch_bounds__(kWG) void k_probe(const Entry *__restrict__ src, int rounds, int stride, const int *__restrict__ jitter,
                                               double *out) {
    __shared__ Entry s_list[kN + 8];
    __shared__ u32x2 s_key[kN + 8];
    __shared__ double s_inv[64];
    __shared__ double s_val[kN + 8];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < kN + 8; i += kWG) {
        s_list[i] = src[i];
        s_key[i] = u32x2{(unsigned)src[i].a, (unsigned)src[i].m};
        s_val[i] = src[i].val;
    }
    if (tid < 64) s_inv[tid] = tid ? 1.0 / tid : 0.0;
    __syncthreads();
    const int p = 1000 + lane;
    double acc = 0.0, acc2 = 0.0;
    unsigned dgrp = 0;
    // per-lane cursor start: uniform stride (V2/V4/V6/V7) or jittered (V3)
    int start = 0;
    if (V == 2 || V == 4 || V == 6 || V == 7) start = (lane & 31) * stride;
    if (V == 3) start = jitter[lane];
    const int T = kN - 32 * stride - 64 > 64 ? kN - 32 * stride - 64 : 64;   // same trip count for every variant
    for (int r = 0; r < rounds; ++r) {
        int k4 = start;
        for (int t = 0; t < T; t += 4) {
            if (V == 0) {
                const Entry e0 = ld16(&s_list[t]), e1 = ld16(&s_list[t + 1]), e2 = ld16(&s_list[t + 2]), e3 = ld16(&s_list[t + 3]);
                acc += ((unsigned)(p - e0.a) < (unsigned)e0.m) ? e0.val : 0.0;
                acc += ((unsigned)(p - e1.a) < (unsigned)e1.m) ? e1.val : 0.0;
                acc += ((unsigned)(p - e2.a) < (unsigned)e2.m) ? e2.val : 0.0;
                acc += ((unsigned)(p - e3.a) < (unsigned)e3.m) ? e3.val : 0.0;
            } else if (V == 1 || V == 6) {
                const int q = V == 1 ? t : (k4 < kN ? k4 : kN);
                const Entry e0 = ld16(&s_list[q]), e1 = ld16(&s_list[q + 1]), e2 = ld16(&s_list[q + 2]), e3 = ld16(&s_list[q + 3]);
                k4 += 4;
                int d;
#define CMPX_ADD(E)                                                                                                   \
    asm volatile("v_sub_u32 %0, %2, %3\n\tv_cmpx_lt_u32 vcc, %0, %4\n\tv_add_f64 %1, %1, %5\n\ts_mov_b64 exec, -1"     \
                 : "=&v"(d), "+v"(acc) : "v"(p), "v"(E.a), "v"(E.m), "v"(E.val) : "vcc")
                CMPX_ADD(e0); CMPX_ADD(e1); CMPX_ADD(e2); CMPX_ADD(e3);
            } else if (V == 2 || V == 3) {
                const int q = k4 < kN ? k4 : kN;
                const Entry e0 = ld16(&s_list[q]), e1 = ld16(&s_list[q + 1]), e2 = ld16(&s_list[q + 2]), e3 = ld16(&s_list[q + 3]);
                k4 += 4;
                acc += ((unsigned)(p - e0.a) < (unsigned)e0.m) ? e0.val : 0.0;
                acc += ((unsigned)(p - e1.a) < (unsigned)e1.m) ? e1.val : 0.0;
                acc += ((unsigned)(p - e2.a) < (unsigned)e2.m) ? e2.val : 0.0;
                acc += ((unsigned)(p - e3.a) < (unsigned)e3.m) ? e3.val : 0.0;
            } else if (V == 4) {
                const int q = k4 < kN ? k4 : kN;
                const u32x2 k0 = s_key[q], k1 = s_key[q + 1], k2 = s_key[q + 2], k3 = s_key[q + 3];
                k4 += 4;
                const double v0 = s_inv[k0.y & 63], v1 = s_inv[k1.y & 63], v2 = s_inv[k2.y & 63], v3 = s_inv[k3.y & 63];
                acc += ((unsigned)(p - (int)k0.x) < k0.y) ? v0 : 0.0;
                acc += ((unsigned)(p - (int)k1.x) < k1.y) ? v1 : 0.0;
                acc += ((unsigned)(p - (int)k2.x) < k2.y) ? v2 : 0.0;
                acc += ((unsigned)(p - (int)k3.x) < k3.y) ? v3 : 0.0;
            } else if (V == 5) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int a = 990 + ((t + u) & 31), m = 25 + ((t + u) & 7);
                    const double val = __longlong_as_double(0x3fa0000000000000ll + (long long)(t + u));
                    acc += ((unsigned)(p - a) < (unsigned)m) ? val : 0.0;
                }
            } else if (V == 8 || V == 9) {
                const Entry e0 = ld16(&s_list[t]), e1 = ld16(&s_list[t + 1]), e2 = ld16(&s_list[t + 2]), e3 = ld16(&s_list[t + 3]);
                if ((t & 31) == 0) dgrp = (unsigned)(p - e0.a);
#define ONE_SEL(E)                                                                                                    \
    {                                                                                                                 \
        const unsigned long long b = (unsigned long long)__double_as_longlong(E.val);                                 \
        const unsigned hi = dgrp < (unsigned)E.m ? (unsigned)(b >> 32) : 0u;                                          \
        const unsigned lo = V == 9 ? (dgrp < (unsigned)E.m ? (unsigned)b : 0u) : (unsigned)b;                         \
        acc += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));                                \
    }
                ONE_SEL(e0) ONE_SEL(e1) ONE_SEL(e2) ONE_SEL(e3)
            } else if (V == 10) {
                const unsigned m0 = s_key[t].y, m1 = s_key[t + 1].y, m2 = s_key[t + 2].y, m3 = s_key[t + 3].y;
                const double v0 = s_val[t], v1 = s_val[t + 1], v2 = s_val[t + 2], v3 = s_val[t + 3];
                if ((t & 31) == 0) dgrp = (unsigned)(p - (int)s_key[t].x);
#define ONE_SEL2(M, VV)                                                                                               \
    {                                                                                                                 \
        const unsigned long long b = (unsigned long long)__double_as_longlong(VV);                                    \
        const unsigned hi = dgrp < M ? (unsigned)(b >> 32) : 0u;                                                      \
        acc += __longlong_as_double((long long)(((unsigned long long)hi << 32) | (unsigned)b));                       \
    }
                ONE_SEL2(m0, v0) ONE_SEL2(m1, v1) ONE_SEL2(m2, v2) ONE_SEL2(m3, v3)
            } else if (V == 11) {
                const u32x4 *g = (const u32x4 *)src + ((t + (r & 1) * 4) & (kN - 1)); // wave-uniform address -> s_load
                const u32x4 q0 = g[0], q1 = g[1], q2 = g[2], q3 = g[3];
#define SEL_S(Q)                                                                                                      \
    acc += ((unsigned)(p - (int)Q.x) < Q.y) ? __longlong_as_double((long long)(((unsigned long long)Q.w << 32) | Q.z)) : 0.0;
                SEL_S(q0) SEL_S(q1) SEL_S(q2) SEL_S(q3)
            } else if (V == 12) {
                const u32x4 *g = (const u32x4 *)src + ((t + (r & 1) * 4) & (kN - 1));
                const u32x4 q0 = g[0], q1 = g[1], q2 = g[2], q3 = g[3];
                int d;
#define CMPX_S(Q)                                                                                                     \
    {                                                                                                                 \
        const unsigned long long vb = ((unsigned long long)Q.w << 32) | Q.z;                                          \
        asm volatile("v_subrev_u32 %0, %3, %2\n\tv_cmpx_gt_u32 vcc, %4, %0\n\tv_add_f64 %1, %1, %5\n\ts_mov_b64 exec, -1" \
                     : "=&v"(d), "+v"(acc) : "v"(p), "s"(Q.x), "s"(Q.y), "s"(vb) : "vcc");                            \
    }
                CMPX_S(q0) CMPX_S(q1) CMPX_S(q2) CMPX_S(q3)
            } else if (V == 7) {
                const int q = k4 < kN ? k4 : kN;
                const Entry e0 = ld16(&s_list[q]), e1 = ld16(&s_list[q + 1]), e2 = ld16(&s_list[q + 2]), e3 = ld16(&s_list[q + 3]);
                k4 += 4;
                acc += ((unsigned)(2 * p - e0.a) < (unsigned)e0.m) ? e0.val : 0.0;
                acc2 += ((unsigned)(2 * p + 1 - e0.a) < (unsigned)e0.m) ? e0.val : 0.0;
                acc += ((unsigned)(2 * p - e1.a) < (unsigned)e1.m) ? e1.val : 0.0;
                acc2 += ((unsigned)(2 * p + 1 - e1.a) < (unsigned)e1.m) ? e1.val : 0.0;
                acc += ((unsigned)(2 * p - e2.a) < (unsigned)e2.m) ? e2.val : 0.0;
                acc2 += ((unsigned)(2 * p + 1 - e2.a) < (unsigned)e2.m) ? e2.val : 0.0;
                acc += ((unsigned)(2 * p - e3.a) < (unsigned)e3.m) ? e3.val : 0.0;
                acc2 += ((unsigned)(2 * p + 1 - e3.a) < (unsigned)e3.m) ? e3.val : 0.0;
            }
        }
    }
    out[(size_t)blockIdx.x * kWG + tid] = acc + acc2;
}

template <int V>
double run(const Entry *d_src, int blocks, int rounds, int stride, const int *d_jit, double *d_out, int &T_out) {
    const int T = kN - 32 * stride - 64 > 64 ? kN - 32 * stride - 64 : 64;
    T_out = T;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_probe<V>), dim3(blocks), dim3(kWG), 0, 0, d_src, 2, stride, d_jit, d_out); // warm-up
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((k_probe<V>), dim3(blocks), dim3(kWG), 0, 0, d_src, rounds, stride, d_jit, d_out);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    std::vector<Entry> h(kN + 8);
    srand(7);
    for (int i = 0; i < kN + 8; ++i) { h[i].a = 960 + (i * 96) / kN + rand() % 3; h[i].m = 25 + rand() % 10; h[i].val = 1.0 / h[i].m; }
    for (int i = kN; i < kN + 8; ++i) h[i].m = 0;
    Entry *d_src; hipMalloc(&d_src, sizeof(Entry) * (kN + 8));
    hipMemcpy(d_src, h.data(), sizeof(Entry) * (kN + 8), hipMemcpyHostToDevice);
    std::vector<int> jit(64);
    for (int stride : {1}) {
        int c = 0;
        for (int l = 0; l < 64; ++l) { if ((l & 31) == 0) c = 0; jit[l] = c; c += rand() % (2 * stride + 1); }
        int *d_jit; hipMalloc(&d_jit, 256);
        hipMemcpy(d_jit, jit.data(), 256, hipMemcpyHostToDevice);
        for (int wgs_per_cu : {1, 2, 4, 6}) {
            const int blocks = 256 * wgs_per_cu;   // one round of residency: wgs_per_cu x 4 waves per CU
            double *d_out; hipMalloc(&d_out, sizeof(double) * blocks * kWG);
            int T;
            const double clk = 2.4e6; // cycles per ms (nominal)
            double ms[13];
            ms[0] = run<0>(d_src, blocks, rounds, stride, d_jit, d_out, T);
            ms[1] = run<1>(d_src, blocks, rounds, stride, d_jit, d_out, T);
            ms[2] = run<2>(d_src, blocks, rounds, stride, d_jit, d_out, T);
            ms[3] = run<3>(d_src, blocks, rounds, stride, d_jit, d_out, T);
            ms[4] = run<4>(d_src, blocks, rounds, stride, d_jit, d_out, T);
            ms[5] = run<5>(d_src, blocks, rounds, stride, d_jit, d_out, T);
            ms[6] = run<6>(d_src, blocks, rounds, stride, d_jit, d_out, T);
            ms[7] = run<7>(d_src, blocks, rounds, stride, d_jit, d_out, T);
            ms[8] = run<8>(d_src, blocks, rounds, stride, d_jit, d_out, T);
            ms[9] = run<9>(d_src, blocks, rounds, stride, d_jit, d_out, T);
            ms[10] = run<10>(d_src, blocks, rounds, stride, d_jit, d_out, T);
            ms[11] = run<11>(d_src, blocks, rounds, stride, d_jit, d_out, T);
            ms[12] = run<12>(d_src, blocks, rounds, stride, d_jit, d_out, T);
            printf("stride %d  waves/SIMD %d  T %d:", stride, wgs_per_cu, T);
            // cycles per list entry per SIMD = ms * clk / (rounds * T * waves per SIMD)
            for (int v = 0; v < 13; ++v) printf("  V%d %.1f", v, ms[v] * clk / ((double)rounds * T * wgs_per_cu));
            printf("   (cycles per entry per SIMD)\n");
            hipFree(d_out);
        }
        hipFree(d_jit);
    }
    return 0;
}
