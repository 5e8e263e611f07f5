// scalar_stream_probe.hip -- the ordered float64 replay of the center rule with the candidates taken through the
// SCALAR path straight from a read-only, position-sorted stream in HBM (round-3 experiment; follow-up of
// replay_probe.hip V12 and scalar_replay_probe.hip, which only measured cache-resident / wave-built lists).
//
// One wave owns 64 (or 128) consecutive positions and walks the 8-byte entries {pos, L} that can reach them; every
// wave streams its own stretch of a large sorted stream (neighbouring waves overlap by a read length, as neighbouring
// chunks do) with s_load_dwordx16, two buffers, the next load in flight while one is replayed.  Per entry:
//     m0 = 2 (L - Lbase) ; m = L - 2 nib ; val = s_movrels_b64 table[m0]           (scalar)
//     t = p' - pos ; v_cmpx (m > t) ; v_add_f64 acc, acc, val ; exec = -1          (3 vector + 1 scalar)
//   A    64 positions per wave                      Aw   + L2 warming (one vector load touches the lines 4 KiB ahead)
//   A2   128 positions per wave (two per lane)      A2w  + L2 warming
//   V    entries by coalesced vector loads, broadcast with v_readlane (5 vector + 3 scalar per entry), no scalar memory
// First measurement (profiles/r03/notes): 16-byte entries {pos + nib, m, 1/m} (no table, 3 vector + 1 scalar) took
// TWICE the time of the 8-byte form -- the loop is bound by bytes through the scalar cache / latency of one 64-byte
// line in flight per wave, not by instructions -- so this version only keeps 8-byte entries.
// Every form is checked bit for bit against a plain per-lane loop over the same entries.
// Reported: SIMD cycles per wave-entry (kernel time x 1024 SIMDs x clock / wave-entries) and ms per 2.1e8 wave-entries
// (what k_center replays on C3).
// build: hipcc --offload-arch=gfx950 -O3 -o scalar_stream_probe scalar_stream_probe.hip ; run: ./scalar_stream_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int kLbase = 22, kTab = 16;   // the SGPR table holds 1/(L - 2 nib) for L in [kLbase, kLbase + kTab)

// fixed registers of the loop: s[36:51] buffer 0, s[52:67] buffer 1, s[68:99] value table, s[30:31] stream pointer,
// s26 groups left, s25 pass counter, s27 saved m0, s[34:35] saved exec, s33 m, s[28:29] value of the entry, s24 scratch
#define SCLOB                                                                                                          \
    "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s33", "s34", "s35", "s36", "s37", "s38", "s39", "s40",    \
        "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", \
        "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", \
        "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", \
        "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "vcc", "scc", "memory"

#define A_ENT(P, L)                                                                                                    \
    "s_lshl1_add_u32 m0, s" #L ", %[c2]\n\t"                                                                            \
    "s_sub_u32 s33, s" #L ", %[n2]\n\t"                                                                                 \
    "s_movrels_b64 s[28:29], s[68:69]\n\t"                                                                              \
    "v_subrev_u32 %[t], s" #P ", %[pn]\n\t"                                                                             \
    "v_cmpx_gt_u32 vcc, s33, %[t]\n\t"                                                                                  \
    "v_add_f64 %[acc], %[acc], s[28:29]\n\t"                                                                            \
    "s_mov_b64 exec, s[34:35]\n\t"
#define A2_ENT(P, L)                                                                                                   \
    "s_lshl1_add_u32 m0, s" #L ", %[c2]\n\t"                                                                            \
    "s_sub_u32 s33, s" #L ", %[n2]\n\t"                                                                                 \
    "s_movrels_b64 s[28:29], s[68:69]\n\t"                                                                              \
    "v_subrev_u32 %[t], s" #P ", %[pn]\n\t"                                                                             \
    "v_cmpx_gt_u32 vcc, s33, %[t]\n\t"                                                                                  \
    "v_add_f64 %[acc], %[acc], s[28:29]\n\t"                                                                            \
    "s_mov_b64 exec, s[34:35]\n\t"                                                                                      \
    "v_subrev_u32 %[t], s" #P ", %[pn2]\n\t"                                                                            \
    "v_cmpx_gt_u32 vcc, s33, %[t]\n\t"                                                                                  \
    "v_add_f64 %[acc2], %[acc2], s[28:29]\n\t"                                                                          \
    "s_mov_b64 exec, s[34:35]\n\t"
#define A_BUF0(E) E(36, 37) E(38, 39) E(40, 41) E(42, 43) E(44, 45) E(46, 47) E(48, 49) E(50, 51)
#define A_BUF1(E) E(52, 53) E(54, 55) E(56, 57) E(58, 59) E(60, 61) E(62, 63) E(64, 65) E(66, 67)

// L2 warming: lane i touches the 64-byte line i of a 4 KiB stretch; at the head the first 8 KiB, then every 32nd pass
// (32 passes x 16 entries x 8 bytes = 4 KiB of progress) the 4 KiB that start 4 KiB ahead.  The loaded word is never
// used and only waited for after the loop.
#define WARM_HEAD                                                                                                      \
    "global_load_dword %[dump], %[voff], s[30:31]\n\t"                                                                  \
    "global_load_dword %[dump], %[voff2], s[30:31]\n\t"
#define WARM_LOOP                                                                                                      \
    "s_add_u32 s25, s25, 1\n\t"                                                                                         \
    "s_and_b32 s24, s25, 31\n\t"                                                                                        \
    "s_cbranch_scc1 3f\n\t"                                                                                             \
    "global_load_dword %[dump], %[voff2], s[30:31]\n"                                                                      \
    "3:\n\t"
#define WARM_TAIL "s_waitcnt vmcnt(0)\n\t"

#define LOOP(ENT, WH, WL, WT)                                                                                          \
    "s_mov_b64 s[34:35], exec\n\t"                                                                                      \
    "s_mov_b64 s[30:31], %[ptr]\n\t"                                                                                    \
    "s_mov_b32 s26, %[n]\n\t"                                                                                           \
    "s_mov_b32 s25, 0\n\t"                                                                                              \
    "s_mov_b32 s27, m0\n\t"                                                                                             \
    "s_load_dwordx16 s[68:83], %[tab], 0x0\n\t"                                                                         \
    "s_load_dwordx16 s[84:99], %[tab], 0x40\n\t"                                                                        \
    "s_load_dwordx16 s[36:51], s[30:31], 0x0\n\t" WH                                                                    \
    "s_waitcnt lgkmcnt(0)\n"                                                                                            \
    "1:\n\t"                                                                                                            \
    "s_load_dwordx16 s[52:67], s[30:31], 0x40\n\t" WL A_BUF0(ENT)                                                       \
    "s_sub_u32 s26, s26, 1\n\t"                                                                                         \
    "s_cmp_eq_u32 s26, 0\n\t"                                                                                           \
    "s_cbranch_scc1 2f\n\t"                                                                                             \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                          \
    "s_load_dwordx16 s[36:51], s[30:31], 0x80\n\t" A_BUF1(ENT)                                                          \
    "s_add_u32 s30, s30, 0x80\n\t"                                                                                      \
    "s_addc_u32 s31, s31, 0\n\t"                                                                                        \
    "s_sub_u32 s26, s26, 1\n\t"                                                                                         \
    "s_cmp_lg_u32 s26, 0\n\t"                                                                                           \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                          \
    "s_cbranch_scc1 1b\n"                                                                                               \
    "2:\n\t"                                                                                                            \
    "s_waitcnt lgkmcnt(0)\n\t" WT                                                                                       \
    "s_mov_b32 m0, s27\n\t"                                                                                             \
    "s_mov_b64 exec, s[34:35]\n\t"

__device__ __forceinline__ unsigned long long uniform64(const void *q) {
    const unsigned long long lp = (unsigned long long)q;
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(lp >> 32)) << 32) |
           (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)lp);
}

// n = number of 64-byte groups (8 entries), >= 1; 8 KiB + 128 bytes behind the last group must be readable
template <bool WARM>
__device__ __forceinline__ double replay_A(double acc, int pn, const void *ptr_, int n_, const void *tab_, int c2_, int n2_, int lane) {
    const unsigned long long ptr = uniform64(ptr_), tab = uniform64(tab_);
    const int n = __builtin_amdgcn_readfirstlane(n_), c2 = __builtin_amdgcn_readfirstlane(c2_), n2 = __builtin_amdgcn_readfirstlane(n2_);
    int t, dump;
    const int voff = lane * 64;
    if (WARM)
        asm volatile(LOOP(A_ENT, WARM_HEAD, WARM_LOOP, WARM_TAIL)
                     : [acc] "+v"(acc), [t] "=&v"(t), [dump] "=&v"(dump)
                     : [pn] "v"(pn), [ptr] "s"(ptr), [n] "s"(n), [tab] "s"(tab), [c2] "s"(c2), [n2] "s"(n2), [voff] "v"(voff), [voff2] "v"(voff + 4096)
                     : SCLOB);
    else
        asm volatile(LOOP(A_ENT, "", "", "")
                     : [acc] "+v"(acc), [t] "=&v"(t)
                     : [pn] "v"(pn), [ptr] "s"(ptr), [n] "s"(n), [tab] "s"(tab), [c2] "s"(c2), [n2] "s"(n2)
                     : SCLOB);
    return acc;
}

template <bool WARM>
__device__ __forceinline__ void replay_A2(double &acc, double &acc2, int pn, const void *ptr_, int n_, const void *tab_, int c2_, int n2_,
                                          int lane) {
    const unsigned long long ptr = uniform64(ptr_), tab = uniform64(tab_);
    const int n = __builtin_amdgcn_readfirstlane(n_), c2 = __builtin_amdgcn_readfirstlane(c2_), n2 = __builtin_amdgcn_readfirstlane(n2_);
    int t, dump;
    const int pn2 = pn + 64, voff = lane * 64;
    if (WARM)
        asm volatile(LOOP(A2_ENT, WARM_HEAD, WARM_LOOP, WARM_TAIL)
                     : [acc] "+v"(acc), [acc2] "+v"(acc2), [t] "=&v"(t), [dump] "=&v"(dump)
                     : [pn] "v"(pn), [pn2] "v"(pn2), [ptr] "s"(ptr), [n] "s"(n), [tab] "s"(tab), [c2] "s"(c2), [n2] "s"(n2), [voff] "v"(voff), [voff2] "v"(voff + 4096)
                     : SCLOB);
    else
        asm volatile(LOOP(A2_ENT, "", "", "")
                     : [acc] "+v"(acc), [acc2] "+v"(acc2), [t] "=&v"(t)
                     : [pn] "v"(pn), [pn2] "v"(pn2), [ptr] "s"(ptr), [n] "s"(n), [tab] "s"(tab), [c2] "s"(c2), [n2] "s"(n2)
                     : SCLOB);
}

struct Job { long long lo; int n; int start; };   // first entry, entries (multiple of 8), first position of the wave

// FORM 0 A, 1 A2, 2 Aw, 3 A2w, 5 reference (plain per-lane loop over the 8-byte stream)
template <int FORM>
__global__ __launch_bounds__(64) void k_replay(const u32x2 *__restrict__ e8, const Job *__restrict__ jobs, const double *__restrict__ tab,
                                               int nib, int two, double *out) {
    const Job jb = jobs[blockIdx.x];
    const int lane = threadIdx.x;
    const int p = jb.start + lane;
    double acc = 0.0, acc2 = 0.0;
    if (jb.n > 0) {
        if (FORM == 0) acc = replay_A<false>(acc, p - nib, e8 + jb.lo, jb.n / 8, tab, -2 * kLbase, 2 * nib, lane);
        else if (FORM == 1) replay_A2<false>(acc, acc2, p - nib, e8 + jb.lo, jb.n / 8, tab, -2 * kLbase, 2 * nib, lane);
        else if (FORM == 2) acc = replay_A<true>(acc, p - nib, e8 + jb.lo, jb.n / 8, tab, -2 * kLbase, 2 * nib, lane);
        else if (FORM == 3) replay_A2<true>(acc, acc2, p - nib, e8 + jb.lo, jb.n / 8, tab, -2 * kLbase, 2 * nib, lane);
        else {
            for (int i = 0; i < jb.n; ++i) {
                const u32x2 e = e8[jb.lo + i];
                const int m = (int)e.y - 2 * nib;
                const double val = tab[(int)e.y - kLbase];
                if ((unsigned)(p - nib - (int)e.x) < (unsigned)m) acc += val;
                if (two && (unsigned)(p + 64 - nib - (int)e.x) < (unsigned)m) acc2 += val;
            }
        }
    }
    out[(size_t)blockIdx.x * 128 + lane] = acc;
    out[(size_t)blockIdx.x * 128 + 64 + lane] = acc2;
}


// ---- form V: the entries of a batch arrive by ONE coalesced vector load (lane j holds entry j); entry j is broadcast
// with two v_readlane, the value comes from the SGPR table (reserved registers s[68:99], loaded once per kernel:
// the kernel is compiled with amdgpu_num_sgpr(64), so the compiler never allocates them):  5 vector + 3 scalar
#define V_ENT(J)                                                                                                       \
    "v_readlane_b32 %[sa], %[va], " #J "\n\t"                                                                           \
    "v_readlane_b32 %[sw], %[vw], " #J "\n\t"                                                                           \
    "s_lshr_b32 m0, %[sw], 16\n\t"                                                                                      \
    "v_subrev_u32 %[t], %[sa], %[p]\n\t"                                                                                \
    "s_movrels_b64 %[val], s[68:69]\n\t"                                                                                \
    "v_cmpx_gt_u16 vcc, %[sw], %[t]\n\t"                                                                                \
    "v_add_f64 %[acc], %[acc], %[val]\n\t"                                                                              \
    "s_mov_b64 exec, -1\n\t"
#define V_ENT8(B) V_ENT(B + 0) V_ENT(B + 1) V_ENT(B + 2) V_ENT(B + 3) V_ENT(B + 4) V_ENT(B + 5) V_ENT(B + 6) V_ENT(B + 7)
#define TABCLOB                                                                                                        \
    "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83",    \
        "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99"

__global__ __attribute__((amdgpu_num_sgpr(64))) __launch_bounds__(64) void k_replay_v(const u32x2 *__restrict__ e8, const Job *__restrict__ jobs,
                                                                                      const double *__restrict__ tab, int nib, double *out) {
    const Job jb = jobs[blockIdx.x];
    const int lane = threadIdx.x;
    const int p = jb.start + lane;
    double acc = 0.0;
    const unsigned long long tp = uniform64(tab);
    asm volatile("s_load_dwordx16 s[68:83], %0, 0x0\n\ts_load_dwordx16 s[84:99], %0, 0x40\n\ts_waitcnt lgkmcnt(0)" ::"s"(tp) : TABCLOB, "memory");
    const long long hi = jb.lo + jb.n, last = hi - 1;
    u32x2 q0 = {0u, 0u}, q1 = q0;
    if (jb.n > 0) {
        q0 = e8[jb.lo + lane < last ? jb.lo + lane : last];
        q1 = e8[jb.lo + 64 + lane < last ? jb.lo + 64 + lane : last];
    }
    auto step = [&](u32x2 &q, long long base) {
        const u32x2 r = q;
        q = e8[base + 128 + lane < last ? base + 128 + lane : last];
        const bool valid = base + lane < hi;
        const int L = (int)r.y, m = L - 2 * nib;
        const int a0 = (int)r.x + nib;
        // entries that cannot touch the wave's positions are neutralised here (the 16-bit compare below needs |p - a0| < 2^15)
        const bool live = valid && m > 0 && a0 < jb.start + 64 && a0 + m > jb.start;
        const unsigned w = (live ? (unsigned)m : 0u) | ((unsigned)(2 * (L - kLbase)) << 16);
        const int a = live ? a0 : jb.start;
        int t, sa, sw;
        double val;
        asm volatile(V_ENT8(0) V_ENT8(8) V_ENT8(16) V_ENT8(24) V_ENT8(32) V_ENT8(40) V_ENT8(48) V_ENT8(56)
                     : [acc] "+v"(acc), [t] "=&v"(t), [sa] "=&s"(sa), [sw] "=&s"(sw), [val] "=&s"(val)
                     : [va] "v"(a), [vw] "v"(w), [p] "v"(p)
                     : "vcc", "scc", "m0");
    };
    for (long long base = jb.lo; base < hi; base += 128) {
        step(q0, base);
        if (base + 64 >= hi) break;
        step(q1, base + 64);
    }
    out[(size_t)blockIdx.x * 128 + lane] = acc;
    out[(size_t)blockIdx.x * 128 + 64 + lane] = 0.0;
}

float run_v(const u32x2 *e8, const Job *jobs, int njobs, const double *tab, int nib, double *out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_replay_v, dim3(njobs), dim3(64), 0, 0, e8, jobs, tab, nib, out);
    hipEventRecord(a, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_replay_v, dim3(njobs), dim3(64), 0, 0, e8, jobs, tab, nib, out);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / 3;
}

// ---- form V4: as V, four entries at a time with the covering masks computed ahead into SGPR pairs (v_cmp, not v_cmpx):
// the dependent chain per entry is then only  s_mov_b64 exec, mask ; v_add_f64  (what a wave that runs alone is bound by)
#define V4_GRP(J0, J1, J2, J3)                                                                                         \
    "v_readlane_b32 %[sa0], %[va], " #J0 "\n\tv_readlane_b32 %[sw0], %[vw], " #J0 "\n\t"                                \
    "v_readlane_b32 %[sa1], %[va], " #J1 "\n\tv_readlane_b32 %[sw1], %[vw], " #J1 "\n\t"                                \
    "v_readlane_b32 %[sa2], %[va], " #J2 "\n\tv_readlane_b32 %[sw2], %[vw], " #J2 "\n\t"                                \
    "v_readlane_b32 %[sa3], %[va], " #J3 "\n\tv_readlane_b32 %[sw3], %[vw], " #J3 "\n\t"                                \
    "s_lshr_b32 m0, %[sw0], 16\n\tv_subrev_u32 %[t0], %[sa0], %[p]\n\ts_movrels_b64 %[val0], s[68:69]\n\t"              \
    "s_lshr_b32 m0, %[sw1], 16\n\tv_subrev_u32 %[t1], %[sa1], %[p]\n\ts_movrels_b64 %[val1], s[68:69]\n\t"              \
    "s_lshr_b32 m0, %[sw2], 16\n\tv_subrev_u32 %[t2], %[sa2], %[p]\n\ts_movrels_b64 %[val2], s[68:69]\n\t"              \
    "s_lshr_b32 m0, %[sw3], 16\n\tv_subrev_u32 %[t3], %[sa3], %[p]\n\ts_movrels_b64 %[val3], s[68:69]\n\t"              \
    "v_cmp_gt_u16 %[mk0], %[sw0], %[t0]\n\tv_cmp_gt_u16 %[mk1], %[sw1], %[t1]\n\t"                                      \
    "v_cmp_gt_u16 %[mk2], %[sw2], %[t2]\n\tv_cmp_gt_u16 %[mk3], %[sw3], %[t3]\n\t"                                      \
    "s_mov_b64 exec, %[mk0]\n\tv_add_f64 %[acc], %[acc], %[val0]\n\t"                                                   \
    "s_mov_b64 exec, %[mk1]\n\tv_add_f64 %[acc], %[acc], %[val1]\n\t"                                                   \
    "s_mov_b64 exec, %[mk2]\n\tv_add_f64 %[acc], %[acc], %[val2]\n\t"                                                   \
    "s_mov_b64 exec, %[mk3]\n\tv_add_f64 %[acc], %[acc], %[val3]\n\t"                                                   \
    "s_mov_b64 exec, -1\n\t"
#define V4_16(B) V4_GRP(B + 0, B + 1, B + 2, B + 3) V4_GRP(B + 4, B + 5, B + 6, B + 7) V4_GRP(B + 8, B + 9, B + 10, B + 11) V4_GRP(B + 12, B + 13, B + 14, B + 15)

__global__ __attribute__((amdgpu_num_sgpr(64))) __launch_bounds__(64) void k_replay_v4(const u32x2 *__restrict__ e8, const Job *__restrict__ jobs,
                                                                                       const double *__restrict__ tab, int nib, double *out) {
    const Job jb = jobs[blockIdx.x];
    const int lane = threadIdx.x;
    const int p = jb.start + lane;
    double acc = 0.0;
    const unsigned long long tp = uniform64(tab);
    asm volatile("s_load_dwordx16 s[68:83], %0, 0x0\n\ts_load_dwordx16 s[84:99], %0, 0x40\n\ts_waitcnt lgkmcnt(0)" ::"s"(tp) : TABCLOB, "memory");
    const long long hi = jb.lo + jb.n, last = hi - 1;
    u32x2 q0 = {0u, 0u}, q1 = q0;
    if (jb.n > 0) {
        q0 = e8[jb.lo + lane < last ? jb.lo + lane : last];
        q1 = e8[jb.lo + 64 + lane < last ? jb.lo + 64 + lane : last];
    }
    auto step = [&](u32x2 &q, long long base) {
        const u32x2 r = q;
        q = e8[base + 128 + lane < last ? base + 128 + lane : last];
        const bool valid = base + lane < hi;
        const int L = (int)r.y, m = L - 2 * nib;
        const int a0 = (int)r.x + nib;
        const bool live = valid && m > 0 && a0 < jb.start + 64 && a0 + m > jb.start;
        const unsigned w = (live ? (unsigned)m : 0u) | ((unsigned)(2 * (L - kLbase)) << 16);
        const int a = live ? a0 : jb.start;
        int t0, t1, t2, t3, sa0, sa1, sa2, sa3, sw0, sw1, sw2, sw3;
        double val0, val1, val2, val3;
        unsigned long long mk0, mk1, mk2, mk3;
        asm volatile(V4_16(0) V4_16(16) V4_16(32) V4_16(48)
                     : [acc] "+v"(acc), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [sa0] "=&s"(sa0), [sa1] "=&s"(sa1),
                       [sa2] "=&s"(sa2), [sa3] "=&s"(sa3), [sw0] "=&s"(sw0), [sw1] "=&s"(sw1), [sw2] "=&s"(sw2), [sw3] "=&s"(sw3),
                       [val0] "=&s"(val0), [val1] "=&s"(val1), [val2] "=&s"(val2), [val3] "=&s"(val3), [mk0] "=&s"(mk0), [mk1] "=&s"(mk1),
                       [mk2] "=&s"(mk2), [mk3] "=&s"(mk3)
                     : [va] "v"(a), [vw] "v"(w), [p] "v"(p)
                     : "vcc", "scc", "m0");
    };
    for (long long base = jb.lo; base < hi; base += 128) {
        step(q0, base);
        if (base + 64 >= hi) break;
        step(q1, base + 64);
    }
    out[(size_t)blockIdx.x * 128 + lane] = acc;
    out[(size_t)blockIdx.x * 128 + 64 + lane] = 0.0;
}

float run_v4(const u32x2 *e8, const Job *jobs, int njobs, const double *tab, int nib, double *out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_replay_v4, dim3(njobs), dim3(64), 0, 0, e8, jobs, tab, nib, out);
    hipEventRecord(a, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_replay_v4, dim3(njobs), dim3(64), 0, 0, e8, jobs, tab, nib, out);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / 3;
}

// ---- forms M*: lane j precomputes the 64-bit mask of the lanes entry j covers; per entry  readlane index, readlane
// mask lo / hi, s_mov m0, s_movrels value, s_mov exec, v_add_f64  (4 vector + 3 scalar).  Table s[42:73], masks
// s[74:89] (two sets of four); the kernels are compiled for 48 SGPRs.
//   Ma  as first built into k_center: index readlanes, then (s_mov m0, mask readlane, movrels) x 4, adds
//   Mb  all twelve readlanes of a group first
//   Mc  Mb software-pipelined: the readlanes and table look-ups of group g+1 are issued before / among the adds of group g
#define M_RL_IDX(S, J0, J1, J2, J3)                                                                                    \
    "v_readlane_b32 %[si" #S "0], %[vix], " #J0 "\n\tv_readlane_b32 %[si" #S "1], %[vix], " #J1 "\n\t"                  \
    "v_readlane_b32 %[si" #S "2], %[vix], " #J2 "\n\tv_readlane_b32 %[si" #S "3], %[vix], " #J3 "\n\t"
#define M_RL_MASK(R0, J0, J1, J2, J3)                                                                                  \
    "v_readlane_b32 s" #R0 "+0, %[vlo], " #J0 "\n\tv_readlane_b32 s" #R0 "+1, %[vhi], " #J0 "\n\t"
#define MASKS_A(J0, J1, J2, J3)                                                                                        \
    "v_readlane_b32 s74, %[vlo], " #J0 "\n\tv_readlane_b32 s75, %[vhi], " #J0 "\n\t"                                    \
    "v_readlane_b32 s76, %[vlo], " #J1 "\n\tv_readlane_b32 s77, %[vhi], " #J1 "\n\t"                                    \
    "v_readlane_b32 s78, %[vlo], " #J2 "\n\tv_readlane_b32 s79, %[vhi], " #J2 "\n\t"                                    \
    "v_readlane_b32 s80, %[vlo], " #J3 "\n\tv_readlane_b32 s81, %[vhi], " #J3 "\n\t"
#define MASKS_B(J0, J1, J2, J3)                                                                                        \
    "v_readlane_b32 s82, %[vlo], " #J0 "\n\tv_readlane_b32 s83, %[vhi], " #J0 "\n\t"                                    \
    "v_readlane_b32 s84, %[vlo], " #J1 "\n\tv_readlane_b32 s85, %[vhi], " #J1 "\n\t"                                    \
    "v_readlane_b32 s86, %[vlo], " #J2 "\n\tv_readlane_b32 s87, %[vhi], " #J2 "\n\t"                                    \
    "v_readlane_b32 s88, %[vlo], " #J3 "\n\tv_readlane_b32 s89, %[vhi], " #J3 "\n\t"
#define LOOKUP(S, K, FILL) "s_mov_b32 m0, %[si" #S #K "]\n\t" FILL "s_movrels_b64 %[val" #S #K "], s[42:43]\n\t"
#define ADD_A(S, K, M) "s_mov_b64 exec, s[" #M "]\n\tv_add_f64 %[acc], %[acc], %[val" #S #K "]\n\t"
// Ma: one group, interleaved as in the first k_center build
#define MA_GRP(J0, J1, J2, J3)                                                                                         \
    M_RL_IDX(a, J0, J1, J2, J3)                                                                                        \
    LOOKUP(a, 0, "v_readlane_b32 s74, %[vlo], " #J0 "\n\t") LOOKUP(a, 1, "v_readlane_b32 s75, %[vhi], " #J0 "\n\t")     \
    LOOKUP(a, 2, "v_readlane_b32 s76, %[vlo], " #J1 "\n\t") LOOKUP(a, 3, "v_readlane_b32 s77, %[vhi], " #J1 "\n\t")     \
    "v_readlane_b32 s78, %[vlo], " #J2 "\n\tv_readlane_b32 s79, %[vhi], " #J2 "\n\t"                                    \
    "v_readlane_b32 s80, %[vlo], " #J3 "\n\tv_readlane_b32 s81, %[vhi], " #J3 "\n\t"                                    \
    ADD_A(a, 0, 74:75) ADD_A(a, 1, 76:77) ADD_A(a, 2, 78:79) ADD_A(a, 3, 80:81) "s_mov_b64 exec, -1\n\t"
// Mb: all readlanes first
#define MB_GRP(J0, J1, J2, J3)                                                                                         \
    M_RL_IDX(a, J0, J1, J2, J3) MASKS_A(J0, J1, J2, J3)                                                                 \
    LOOKUP(a, 0, "s_nop 0\n\t") LOOKUP(a, 1, "s_nop 0\n\t") LOOKUP(a, 2, "s_nop 0\n\t") LOOKUP(a, 3, "s_nop 0\n\t")     \
    ADD_A(a, 0, 74:75) ADD_A(a, 1, 76:77) ADD_A(a, 2, 78:79) ADD_A(a, 3, 80:81) "s_mov_b64 exec, -1\n\t"
// Mc: pipelined pair of groups (A then B): B's readlanes before A's adds, B's look-ups among A's adds
#define MC_PRO(J0, J1, J2, J3)                                                                                         \
    M_RL_IDX(a, J0, J1, J2, J3) MASKS_A(J0, J1, J2, J3)                                                                 \
    LOOKUP(a, 0, "s_nop 0\n\t") LOOKUP(a, 1, "s_nop 0\n\t") LOOKUP(a, 2, "s_nop 0\n\t") LOOKUP(a, 3, "s_nop 0\n\t")
// adds of set X (masks MX) with the read-ahead of set Y (group J*) and its look-ups in between
#define MC_STEP_AB(J0, J1, J2, J3)                                                                                     \
    M_RL_IDX(b, J0, J1, J2, J3) MASKS_B(J0, J1, J2, J3)                                                                 \
    "s_mov_b64 exec, s[74:75]\n\tv_add_f64 %[acc], %[acc], %[vala0]\n\ts_mov_b32 m0, %[sib0]\n\t"                        \
    "s_mov_b64 exec, s[76:77]\n\ts_movrels_b64 %[valb0], s[42:43]\n\tv_add_f64 %[acc], %[acc], %[vala1]\n\ts_mov_b32 m0, %[sib1]\n\t" \
    "s_mov_b64 exec, s[78:79]\n\ts_movrels_b64 %[valb1], s[42:43]\n\tv_add_f64 %[acc], %[acc], %[vala2]\n\ts_mov_b32 m0, %[sib2]\n\t" \
    "s_mov_b64 exec, s[80:81]\n\ts_movrels_b64 %[valb2], s[42:43]\n\tv_add_f64 %[acc], %[acc], %[vala3]\n\ts_mov_b32 m0, %[sib3]\n\t" \
    "s_mov_b64 exec, -1\n\ts_movrels_b64 %[valb3], s[42:43]\n\t"
#define MC_STEP_BA(J0, J1, J2, J3)                                                                                     \
    M_RL_IDX(a, J0, J1, J2, J3) MASKS_A(J0, J1, J2, J3)                                                                 \
    "s_mov_b64 exec, s[82:83]\n\tv_add_f64 %[acc], %[acc], %[valb0]\n\ts_mov_b32 m0, %[sia0]\n\t"                        \
    "s_mov_b64 exec, s[84:85]\n\ts_movrels_b64 %[vala0], s[42:43]\n\tv_add_f64 %[acc], %[acc], %[valb1]\n\ts_mov_b32 m0, %[sia1]\n\t" \
    "s_mov_b64 exec, s[86:87]\n\ts_movrels_b64 %[vala1], s[42:43]\n\tv_add_f64 %[acc], %[acc], %[valb2]\n\ts_mov_b32 m0, %[sia2]\n\t" \
    "s_mov_b64 exec, s[88:89]\n\ts_movrels_b64 %[vala2], s[42:43]\n\tv_add_f64 %[acc], %[acc], %[valb3]\n\ts_mov_b32 m0, %[sia3]\n\t" \
    "s_mov_b64 exec, -1\n\ts_movrels_b64 %[vala3], s[42:43]\n\t"
#define MC_EPI_B                                                                                                       \
    ADD_A(b, 0, 82:83) ADD_A(b, 1, 84:85) ADD_A(b, 2, 86:87) ADD_A(b, 3, 88:89) "s_mov_b64 exec, -1\n\t"
#define GRP16(G, B) G(B + 0, B + 1, B + 2, B + 3) G(B + 4, B + 5, B + 6, B + 7) G(B + 8, B + 9, B + 10, B + 11) G(B + 12, B + 13, B + 14, B + 15)
#define MC_PAIR(B) MC_STEP_AB(B + 4, B + 5, B + 6, B + 7) MC_STEP_BA(B + 8, B + 9, B + 10, B + 11)
#define MC_64                                                                                                          \
    MC_PRO(0, 1, 2, 3) MC_PAIR(0) MC_PAIR(8) MC_PAIR(16) MC_PAIR(24) MC_PAIR(32) MC_PAIR(40) MC_PAIR(48)                \
    MC_STEP_AB(60, 61, 62, 63) MC_EPI_B
#define MTABCLOB                                                                                                       \
    "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57",    \
        "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73"
#define MMASKCLOB "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89"

template <int FORM>
__global__ __attribute__((amdgpu_num_sgpr(48))) __launch_bounds__(64) void k_replay_m(const u32x2 *__restrict__ e8, const Job *__restrict__ jobs,
                                                                                      const double *__restrict__ tab, int nib, double *out) {
    const Job jb = jobs[blockIdx.x];
    const int lane = threadIdx.x;
    double acc = 0.0;
    const unsigned long long tp = uniform64(tab);
    asm volatile("s_load_dwordx2 s[42:43], %0, 0x0\n\ts_load_dwordx4 s[44:47], %0, 0x8\n\ts_load_dwordx16 s[48:63], %0, 0x18\n\t"
                 "s_load_dwordx8 s[64:71], %0, 0x58\n\ts_load_dwordx2 s[72:73], %0, 0x78\n\ts_waitcnt lgkmcnt(0)" ::"s"(tp) : MTABCLOB, "memory");
    const long long hi = jb.lo + jb.n, last = hi - 1;
    u32x2 q0 = {0u, 0u}, q1 = q0;
    if (jb.n > 0) {
        q0 = e8[jb.lo + lane < last ? jb.lo + lane : last];
        q1 = e8[jb.lo + 64 + lane < last ? jb.lo + 64 + lane : last];
    }
    auto step = [&](u32x2 &q, long long base) {
        const u32x2 r = q;
        q = e8[base + 128 + lane < last ? base + 128 + lane : last];
        const bool valid = base + lane < hi;
        const int L = (int)r.y, m = L - 2 * nib;
        const int rel = (int)r.x + nib - jb.start, rlo = rel > 0 ? rel : 0, rhi = rel + m < 64 ? rel + m : 64, nl = rhi - rlo;
        const unsigned long long mask = (valid && m > 0 && nl > 0 && rel < 64) ? ((nl >= 64 ? ~0ull : ((1ull << nl) - 1ull)) << rlo) : 0ull;
        const unsigned mlo = (unsigned)mask, mhi = (unsigned)(mask >> 32), vix = (unsigned)(2 * (L - kLbase));
        int sia0, sia1, sia2, sia3, sib0, sib1, sib2, sib3, m0s;
        double vala0, vala1, vala2, vala3, valb0, valb1, valb2, valb3;
#define M_OPERANDS                                                                                                     \
    : [acc] "+v"(acc), [sia0] "=&s"(sia0), [sia1] "=&s"(sia1), [sia2] "=&s"(sia2), [sia3] "=&s"(sia3), [sib0] "=&s"(sib0),  \
      [sib1] "=&s"(sib1), [sib2] "=&s"(sib2), [sib3] "=&s"(sib3), [vala0] "=&s"(vala0), [vala1] "=&s"(vala1),          \
      [vala2] "=&s"(vala2), [vala3] "=&s"(vala3), [valb0] "=&s"(valb0), [valb1] "=&s"(valb1), [valb2] "=&s"(valb2),    \
      [valb3] "=&s"(valb3), [m0s] "=&s"(m0s)                                                                           \
    : [vlo] "v"(mlo), [vhi] "v"(mhi), [vix] "v"(vix)                                                                   \
    : MMASKCLOB, "scc"
        if (FORM == 0) asm volatile("s_mov_b32 %[m0s], m0\n\t" GRP16(MA_GRP, 0) GRP16(MA_GRP, 16) GRP16(MA_GRP, 32) GRP16(MA_GRP, 48) "s_mov_b32 m0, %[m0s]\n\t" M_OPERANDS);
        else if (FORM == 1) asm volatile("s_mov_b32 %[m0s], m0\n\t" GRP16(MB_GRP, 0) GRP16(MB_GRP, 16) GRP16(MB_GRP, 32) GRP16(MB_GRP, 48) "s_mov_b32 m0, %[m0s]\n\t" M_OPERANDS);
        else asm volatile("s_mov_b32 %[m0s], m0\n\t" MC_64 "s_mov_b32 m0, %[m0s]\n\t" M_OPERANDS);
    };
    for (long long base = jb.lo; base < hi; base += 128) {
        step(q0, base);
        if (base + 64 >= hi) break;
        step(q1, base + 64);
    }
    out[(size_t)blockIdx.x * 128 + lane] = acc;
    out[(size_t)blockIdx.x * 128 + 64 + lane] = 0.0;
}

template <int FORM>
float run_m(const u32x2 *e8, const Job *jobs, int njobs, const double *tab, int nib, double *out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_replay_m<FORM>), dim3(njobs), dim3(64), 0, 0, e8, jobs, tab, nib, out);
    hipEventRecord(a, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k_replay_m<FORM>), dim3(njobs), dim3(64), 0, 0, e8, jobs, tab, nib, out);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / 3;
}

template <int FORM>
float run(const u32x2 *e8, const Job *jobs, int njobs, const double *tab, int nib, int two, double *out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_replay<FORM>), dim3(njobs), dim3(64), 0, 0, e8, jobs, tab, nib, two, out);
    hipEventRecord(a, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k_replay<FORM>), dim3(njobs), dim3(64), 0, 0, e8, jobs, tab, nib, two, out);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / 3;
}

int main(int argc, char **argv) {
    const long long npos = argc > 1 ? atoll(argv[1]) : 12000000;   // positions of the one strand modelled here
    const double sigma = argc > 2 ? atof(argv[2]) : 1.2;           // log-normal spread of the expression per 2 kb block
    const int nib = 0;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const double clk = prop.clockRate * 1e3;   // Hz
    printf("device %s, %d CUs, %.0f MHz\n", prop.name, prop.multiProcessorCount, clk / 1e6);
    std::vector<double> tab(kTab);
    for (int j = 0; j < kTab; ++j) { const int m = kLbase + j - 2 * nib; tab[j] = m > 0 ? 1.0 / m : 0.0; }
    double *d_tab; hipMalloc(&d_tab, sizeof(double) * kTab);
    hipMemcpy(d_tab, tab.data(), sizeof(double) * kTab, hipMemcpyHostToDevice);
    for (double rho : {0.5, 4.0, 8.0}) {
        // sorted stream: expression varies along the strand (log-normal per 2 kb block), read lengths 25..34
        std::mt19937_64 rng(12345);
        std::lognormal_distribution<double> expr(0.0, sigma);
        const double mean = std::exp(sigma * sigma / 2);
        std::vector<u32x2> e8;
        e8.reserve((size_t)(npos * rho * 1.3));
        double w = 1.0;
        for (long long pos = 0; pos < npos; ++pos) {
            if (pos % 2048 == 0) w = expr(rng) / mean;
            std::poisson_distribution<int> cnt(rho * w);
            const int c = cnt(rng);
            for (int k = 0; k < c; ++k) e8.push_back(u32x2{(unsigned)pos, (unsigned)(25 + rng() % 10)});
        }
        const long long ne = (long long)e8.size();
        for (int k = 0; k < 2048 + 64; ++k) e8.push_back(u32x2{0x7fffff00u, 25u});   // readable padding that covers nothing
        u32x2 *d_e8;
        hipMalloc(&d_e8, sizeof(u32x2) * e8.size());
        hipMemcpy(d_e8, e8.data(), sizeof(u32x2) * e8.size(), hipMemcpyHostToDevice);
        for (int width : {64, 128}) {
            // jobs: wave j owns positions [j * width, +width); its entries start in [start - 33, start + width)
            std::vector<Job> jobs;
            long long entries = 0;
            {
                long long lo = 0, hi = 0;
                for (long long s = 0; s + width <= npos; s += width) {
                    while (lo < ne && (long long)e8[(size_t)lo].x < s - 33) ++lo;
                    while (hi < ne && (long long)e8[(size_t)hi].x < s + width) ++hi;
                    long long l8 = lo & ~7ll;                             // 64-byte aligned start (surplus entries cover nothing)
                    const long long n = ((hi - l8) + 7) / 8 * 8;
                    jobs.push_back(Job{l8, hi > lo ? (int)n : 0, (int)s});
                    entries += hi > lo ? n : 0;
                }
            }
            // heavy first, as the dispatch list does
            std::stable_sort(jobs.begin(), jobs.end(), [](const Job &a, const Job &b) { return a.n > b.n; });
            Job *d_jobs; hipMalloc(&d_jobs, sizeof(Job) * jobs.size());
            hipMemcpy(d_jobs, jobs.data(), sizeof(Job) * jobs.size(), hipMemcpyHostToDevice);
            const int nj = (int)jobs.size();
            double *d_out, *d_ref; hipMalloc(&d_out, sizeof(double) * 128 * (size_t)nj); hipMalloc(&d_ref, sizeof(double) * 128 * (size_t)nj);
            const int two = width == 128;
            const float ms_ref = run<5>(d_e8, d_jobs, nj, d_tab, nib, two, d_ref);
            std::vector<double> ref(128 * (size_t)nj), got(128 * (size_t)nj);
            hipMemcpy(ref.data(), d_ref, sizeof(double) * ref.size(), hipMemcpyDeviceToHost);
            printf("rho %.1f width %d: %lld entries (%.0f MB), %d waves, %lld wave-entries (max %d per wave = %.3f ms at 16 cycles each); plain per-lane loop %.3f ms\n",
                   rho, width, ne, ne * 8 / 1e6, nj, entries, jobs[0].n, jobs[0].n * 16.0 / clk * 1e3, ms_ref);
            auto report = [&](const char *name, float ms) {
                hipMemcpy(got.data(), d_out, sizeof(double) * got.size(), hipMemcpyDeviceToHost);
                long long bad = 0;
                for (size_t i = 0; i < got.size(); ++i) bad += memcmp(&got[i], &ref[i], 8) != 0;
                printf("   %-4s %.3f ms  %.2f SIMD-cycles per wave-entry  (%.2f per 64-position entry; 2.1e8 of those = %.2f ms)  %lld values differ\n", name, ms,
                       ms * 1e-3 * clk * 1024.0 / (double)entries, ms * 1e-3 * clk * 1024.0 / (double)entries / (two ? 2.0 : 1.0),
                       ms / (double)entries / (two ? 2.0 : 1.0) * 2.1e8, bad);
                hipMemset(d_out, 0, sizeof(double) * got.size());
            };
            if (!two) {
                report("A", run<0>(d_e8, d_jobs, nj, d_tab, nib, two, d_out));
                report("Aw", run<2>(d_e8, d_jobs, nj, d_tab, nib, two, d_out));
                report("V", run_v(d_e8, d_jobs, nj, d_tab, nib, d_out));
                report("V4", run_v4(d_e8, d_jobs, nj, d_tab, nib, d_out));
                report("Ma", run_m<0>(d_e8, d_jobs, nj, d_tab, nib, d_out));
                report("Mb", run_m<1>(d_e8, d_jobs, nj, d_tab, nib, d_out));
                report("Mc", run_m<2>(d_e8, d_jobs, nj, d_tab, nib, d_out));
                if (rho == 4.0) {   // the heaviest wave alone: what one wave's dependent chain costs per entry
                    report("V1w", run_v(d_e8, d_jobs, 1, d_tab, nib, d_out));
                    printf("        (one wave, %d entries: V %.1f, ", jobs[0].n, run_v(d_e8, d_jobs, 1, d_tab, nib, d_out) * 1e-3 * clk / jobs[0].n);
                    printf("V4 %.1f, ", run_v4(d_e8, d_jobs, 1, d_tab, nib, d_out) * 1e-3 * clk / jobs[0].n);
                    printf("Ma %.1f, ", run_m<0>(d_e8, d_jobs, 1, d_tab, nib, d_out) * 1e-3 * clk / jobs[0].n);
                    printf("Mb %.1f, ", run_m<1>(d_e8, d_jobs, 1, d_tab, nib, d_out) * 1e-3 * clk / jobs[0].n);
                    printf("Mc %.1f, ", run_m<2>(d_e8, d_jobs, 1, d_tab, nib, d_out) * 1e-3 * clk / jobs[0].n);
                    printf("A %.1f, Aw %.1f cycles per entry)\n", run<0>(d_e8, d_jobs, 1, d_tab, nib, two, d_out) * 1e-3 * clk / jobs[0].n,
                           run<2>(d_e8, d_jobs, 1, d_tab, nib, two, d_out) * 1e-3 * clk / jobs[0].n);
                }
            } else {
                report("A2", run<1>(d_e8, d_jobs, nj, d_tab, nib, two, d_out));
                report("A2w", run<3>(d_e8, d_jobs, nj, d_tab, nib, two, d_out));
            }
            hipFree(d_jobs); hipFree(d_out); hipFree(d_ref);
        }
        hipFree(d_e8);
    }
    return 0;
}
