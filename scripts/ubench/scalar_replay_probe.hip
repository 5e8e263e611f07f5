// scalar_replay_probe.hip -- the ordered float64 replay with the covering test moved out of the loop
// (scratch experiment, follow-up of replay_probe.hip).
//
// replay_probe.hip showed every form that tests "does this entry cover my position" per lane and entry
// at 18 - 23 cycles per wave and entry.  Here the entry carries the ANSWER: {64-bit lane mask, 1/m}.  The
// replay of one entry is then one scalar and one vector instruction,
//     s_mov_b64 exec, mask ; v_add_f64 acc, acc, val
// with the entries arriving through the scalar cache (s_load_dwordx16 = 4 entries).  Lanes outside the
// mask keep their accumulator, so this is the reference's conditional add, bit for bit.
//   K0  every wave replays the same list (scalar-cache resident)
//   K1  every wave writes its own 64-entry list with vector stores (lane i -> entry i), invalidates the
//       scalar cache and replays it -- the production pattern (list built by the wave's compaction
//       step) -- and checks the result against a select-form replay of the same entries.
// Reported: cycles per wave and entry.
// build: hipcc --offload-arch=gfx950 -O3 -o scalar_replay_probe scalar_replay_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kWG = 256;

// replay `ngroups8` x 8 entries of 16 bytes at `list` (64-byte aligned; one more 64-byte line behind the
// end must be readable) into acc.  s[40:79] are used as the entry buffers and loop state.
__device__ __forceinline__ double replay_masked(double acc, const void *list_, int ngroups8_) {
    // wave-uniform by construction; tell the compiler (the operands must sit in SGPRs)
    const unsigned long long lp = (unsigned long long)list_;
    const unsigned long long list = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(lp >> 32)) << 32) |
                                    (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)lp);
    const int ngroups8 = __builtin_amdgcn_readfirstlane(ngroups8_);
#define PROC(B0, B1, B2, B3, B4, B5, B6, B7)                                                                          \
    "s_mov_b64 exec, s[" #B0 ":" #B1 "]\n\tv_add_f64 %[acc], %[acc], s[" #B2 ":" #B3 "]\n\t"                            \
    "s_mov_b64 exec, s[" #B4 ":" #B5 "]\n\tv_add_f64 %[acc], %[acc], s[" #B6 ":" #B7 "]\n\t"
    asm volatile(
        "s_mov_b64 s[76:77], exec\n\t"
        "s_mov_b64 s[74:75], %[ptr]\n\t"
        "s_mov_b32 s72, %[n]\n\t"
        "s_load_dwordx16 s[40:55], s[74:75], 0x0\n\t"
        "s_waitcnt lgkmcnt(0)\n"
        "1:\n\t"
        "s_load_dwordx16 s[56:71], s[74:75], 0x40\n\t"
        PROC(40, 41, 42, 43, 44, 45, 46, 47) PROC(48, 49, 50, 51, 52, 53, 54, 55)
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_load_dwordx16 s[40:55], s[74:75], 0x80\n\t"
        PROC(56, 57, 58, 59, 60, 61, 62, 63) PROC(64, 65, 66, 67, 68, 69, 70, 71)
        "s_add_u32 s74, s74, 0x80\n\t"
        "s_addc_u32 s75, s75, 0\n\t"
        "s_sub_u32 s72, s72, 1\n\t"
        "s_cmp_lg_u32 s72, 0\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_mov_b64 exec, s[76:77]\n\t"
        : [acc] "+v"(acc)
        : [ptr] "s"(list), [n] "s"(ngroups8)
        : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55",
          "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71",
          "s72", "s74", "s75", "s76", "s77", "scc", "memory");
#undef PROC
    return acc;
}

__global__ __launch_bounds__(kWG) void k_hot(const u32x4 *__restrict__ list, int n_entries, int rounds, double *out) {
    double acc = 0.0;
    for (int r = 0; r < rounds; ++r) acc = replay_masked(acc, list, n_entries / 8);
    out[(size_t)blockIdx.x * kWG + threadIdx.x] = acc;
}

// lane i of every wave builds entry i: positions [a, a+m) of the wave's 64, value 1/m
__global__ __launch_bounds__(kWG) void k_own(u32x4 *scratch, int rounds, double *out, int *mismatches, int check) {
    const int lane = threadIdx.x & 63, wave = (int)((blockIdx.x * kWG + threadIdx.x) >> 6);
    u32x4 *slot = scratch + (size_t)wave * 72;            // 64 entries + one padding line (+ spare)
    double acc = 0.0, ref = 0.0;
    unsigned seed = 12345u + 977u * (unsigned)wave + (unsigned)lane;
    if (lane < 8) slot[64 + lane] = u32x4{0u, 0u, 0u, 0u};
    for (int r = 0; r < rounds; ++r) {
        seed = seed * 1664525u + 1013904223u;
        const int a = (int)((seed >> 8) % 90u) - 26, m = 25 + (int)((seed >> 20) % 11u);
        const int lo = a < 0 ? 0 : a, hi = a + m > 64 ? 64 : a + m;
        const unsigned long long mask = hi > lo ? (((hi - lo) == 64 ? ~0ull : ((1ull << (hi - lo)) - 1ull)) << lo) : 0ull;
        const double val = 1.0 / (double)m;
        const unsigned long long vb = (unsigned long long)__double_as_longlong(val);
        slot[lane] = u32x4{(unsigned)mask, (unsigned)(mask >> 32), (unsigned)vb, (unsigned)(vb >> 32)};
        asm volatile("s_waitcnt vmcnt(0)\n\ts_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        acc = replay_masked(acc, slot, 8);
        if (check) {
            for (int i = 0; i < 64; ++i) {
                const unsigned long long mi = __shfl(mask, i);
                const double vi = __shfl(val, i);
                if ((mi >> lane) & 1ull) ref += vi;
            }
        }
    }
    if (check && __double_as_longlong(acc) != __double_as_longlong(ref)) atomicAdd(mismatches, 1);
    out[(size_t)blockIdx.x * kWG + threadIdx.x] = acc;
}


// as replay_masked with two 64-byte lines per buffer (16 entries per loop iteration; `ngroups16` iterations; two
// more lines behind the end must be readable).  GLC: every load bypasses the scalar cache (no invalidate needed).
template <bool GLC>
__device__ __forceinline__ double replay_masked2(double acc, const void *list_, int ngroups16_) {
    const unsigned long long lp = (unsigned long long)list_;
    const unsigned long long list = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(lp >> 32)) << 32) |
                                    (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)lp);
    const int ngroups16 = __builtin_amdgcn_readfirstlane(ngroups16_);
#define PROC4(B)                                                                                                       \
    "s_mov_b64 exec, s[" #B "+0:" #B "+1]\n\tv_add_f64 %[acc], %[acc], s[" #B "+2:" #B "+3]\n\t"                         \
    "s_mov_b64 exec, s[" #B "+4:" #B "+5]\n\tv_add_f64 %[acc], %[acc], s[" #B "+6:" #B "+7]\n\t"                         \
    "s_mov_b64 exec, s[" #B "+8:" #B "+9]\n\tv_add_f64 %[acc], %[acc], s[" #B "+10:" #B "+11]\n\t"                       \
    "s_mov_b64 exec, s[" #B "+12:" #B "+13]\n\tv_add_f64 %[acc], %[acc], s[" #B "+14:" #B "+15]\n\t"
#define LOADS(G)                                                                                                       \
    asm volatile(                                                                                                      \
        "s_mov_b64 s[30:31], exec\n\t"                                                                                 \
        "s_mov_b64 s[32:33], %[ptr]\n\t"                                                                               \
        "s_mov_b32 s34, %[n]\n\t"                                                                                      \
        "s_load_dwordx16 s[36:51], s[32:33], 0x0" G "\n\t"                                                             \
        "s_load_dwordx16 s[52:67], s[32:33], 0x40" G "\n\t"                                                            \
        "s_waitcnt lgkmcnt(0)\n"                                                                                       \
        "1:\n\t"                                                                                                       \
        "s_load_dwordx16 s[68:83], s[32:33], 0x80" G "\n\t"                                                            \
        "s_load_dwordx16 s[84:99], s[32:33], 0xc0" G "\n\t"                                                            \
        PROC4(36) PROC4(52)                                                                                            \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                     \
        "s_load_dwordx16 s[36:51], s[32:33], 0x100" G "\n\t"                                                           \
        "s_load_dwordx16 s[52:67], s[32:33], 0x140" G "\n\t"                                                           \
        PROC4(68) PROC4(84)                                                                                            \
        "s_add_u32 s32, s32, 0x100\n\t"                                                                                \
        "s_addc_u32 s33, s33, 0\n\t"                                                                                   \
        "s_sub_u32 s34, s34, 1\n\t"                                                                                    \
        "s_cmp_lg_u32 s34, 0\n\t"                                                                                      \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                     \
        "s_cbranch_scc1 1b\n\t"                                                                                        \
        "s_mov_b64 exec, s[30:31]\n\t"                                                                                 \
        : [acc] "+v"(acc)                                                                                              \
        : [ptr] "s"(list), [n] "s"(ngroups16)                                                                          \
        : "s30", "s31", "s32", "s33", "s34", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47",  \
          "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64",  \
          "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81",  \
          "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98",  \
          "s99", "scc", "memory")
    if (GLC) LOADS(" glc"); else LOADS("");
#undef LOADS
#undef PROC4
    return acc;
}

// the production pattern at ring size R (entries per flush): lanes write R entries (R/64 stores per lane),
// then the wave replays them.  MODE 0: replay_masked + s_dcache_inv; 1: replay_masked2 + s_dcache_inv; 2: replay_masked2 glc
template <int R, int MODE>
__global__ __launch_bounds__(kWG) void k_ring(u32x4 *scratch, int rounds, double *out, int *mismatches, int check) {
    const int lane = threadIdx.x & 63, wave = (int)((blockIdx.x * kWG + threadIdx.x) >> 6);
    u32x4 *slot = scratch + (size_t)wave * (R + 16);
    double acc = 0.0, ref = 0.0;
    unsigned seed = 12345u + 977u * (unsigned)wave + (unsigned)lane;
    if (lane < 16) slot[R + lane] = u32x4{0u, 0u, 0u, 0u};
    for (int r = 0; r < rounds; ++r) {
        for (int b = 0; b < R / 64; ++b) {
            seed = seed * 1664525u + 1013904223u;
            const int a = (int)((seed >> 8) % 90u) - 26, m = 25 + (int)((seed >> 20) % 11u);
            const int lo = a < 0 ? 0 : a, hi = a + m > 64 ? 64 : a + m;
            const unsigned long long mask = hi > lo ? (((hi - lo) == 64 ? ~0ull : ((1ull << (hi - lo)) - 1ull)) << lo) : 0ull;
            const double val = 1.0 / (double)m;
            const unsigned long long vb = (unsigned long long)__double_as_longlong(val);
            slot[b * 64 + lane] = u32x4{(unsigned)mask, (unsigned)(mask >> 32), (unsigned)vb, (unsigned)(vb >> 32)};
            if (check) {
                for (int i = 0; i < 64; ++i) {
                    const unsigned long long mi = __shfl(mask, i);
                    const double vi = __shfl(val, i);
                    if ((mi >> lane) & 1ull) ref += vi;
                }
            }
        }
        if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        if (MODE == 0) acc = replay_masked(acc, slot, R / 8);
        else acc = replay_masked2<MODE == 2>(acc, slot, R / 16);
    }
    if (check && __double_as_longlong(acc) != __double_as_longlong(ref)) atomicAdd(mismatches, 1);
    out[(size_t)blockIdx.x * kWG + threadIdx.x] = acc;
}

template <int R, int MODE>
void run_ring(int blocks, int rounds, int wgs_per_cu, double *d_out, int *d_mis) {
    u32x4 *d_scratch; hipMalloc(&d_scratch, sizeof(u32x4) * (R + 16) * (size_t)blocks * 4);
    hipMemset(d_scratch, 0, sizeof(u32x4) * (R + 16) * (size_t)blocks * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms;
    hipMemset(d_mis, 0, 4);
    hipLaunchKernelGGL((k_ring<R, MODE>), dim3(blocks), dim3(kWG), 0, 0, d_scratch, 20, d_out, d_mis, 1);
    int mis = -1; hipMemcpy(&mis, d_mis, 4, hipMemcpyDeviceToHost);
    const int rr = rounds * 64 / R;
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((k_ring<R, MODE>), dim3(blocks), dim3(kWG), 0, 0, d_scratch, rr, d_out, d_mis, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    printf("   ring %3d mode %d: %.2f cycles per wave and entry, %d lanes differ\n", R, MODE, ms * 2.4e6 / ((double)rr * R * wgs_per_cu), mis);
    hipFree(d_scratch);
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    const int n_entries = 512;
    std::vector<u32x4> h(n_entries + 8);
    srand(7);
    for (int i = 0; i < n_entries + 8; ++i) {
        const int a = rand() % 90 - 26, m = 25 + rand() % 11;
        const int lo = a < 0 ? 0 : a, hi = a + m > 64 ? 64 : a + m;
        const unsigned long long mask = (i < n_entries && hi > lo) ? (((1ull << (hi - lo)) - 1ull) << lo) : 0ull;
        const double val = 1.0 / m;
        unsigned long long vb;
        memcpy(&vb, &val, 8);
        h[i] = u32x4{(unsigned)mask, (unsigned)(mask >> 32), (unsigned)vb, (unsigned)(vb >> 32)};
    }
    u32x4 *d_list; hipMalloc(&d_list, sizeof(u32x4) * h.size());
    hipMemcpy(d_list, h.data(), sizeof(u32x4) * h.size(), hipMemcpyHostToDevice);
    int *d_mis; hipMalloc(&d_mis, 4);
    const double clk = 2.4e6;
    for (int wgs_per_cu : {1, 2, 4, 8}) {
        const int blocks = 256 * wgs_per_cu;
        double *d_out; hipMalloc(&d_out, sizeof(double) * blocks * kWG);
        u32x4 *d_scratch; hipMalloc(&d_scratch, sizeof(u32x4) * 72 * (size_t)blocks * 4);
        hipMemset(d_scratch, 0, sizeof(u32x4) * 72 * (size_t)blocks * 4);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float ms0, ms1;
        hipLaunchKernelGGL(k_hot, dim3(blocks), dim3(kWG), 0, 0, d_list, n_entries, 2, d_out);
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(k_hot, dim3(blocks), dim3(kWG), 0, 0, d_list, n_entries, rounds / 8, d_out);
        hipEventRecord(b, 0); hipEventSynchronize(b); hipEventElapsedTime(&ms0, a, b);
        hipMemset(d_mis, 0, 4);
        hipLaunchKernelGGL(k_own, dim3(blocks), dim3(kWG), 0, 0, d_scratch, 50, d_out, d_mis, 1);
        int mis = -1; hipMemcpy(&mis, d_mis, 4, hipMemcpyDeviceToHost);
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(k_own, dim3(blocks), dim3(kWG), 0, 0, d_scratch, rounds, d_out, d_mis, 0);
        hipEventRecord(b, 0); hipEventSynchronize(b); hipEventElapsedTime(&ms1, a, b);
        printf("waves/SIMD %d: K0 %.2f cycles per wave and entry (shared list); K1 %.2f (own list: 64 stores + invalidate + replay), "
               "%d lanes differ from the select form\n", wgs_per_cu, ms0 * clk / ((double)(rounds / 8) * n_entries * wgs_per_cu),
               ms1 * clk / ((double)rounds * 64 * wgs_per_cu), mis);
        run_ring<64, 0>(blocks, rounds, wgs_per_cu, d_out, d_mis);
        run_ring<64, 1>(blocks, rounds, wgs_per_cu, d_out, d_mis);
        run_ring<256, 0>(blocks, rounds, wgs_per_cu, d_out, d_mis);
        run_ring<256, 1>(blocks, rounds, wgs_per_cu, d_out, d_mis);
        run_ring<256, 2>(blocks, rounds, wgs_per_cu, d_out, d_mis);
        hipFree(d_out); hipFree(d_scratch);
    }
    return 0;
}
