// stage_kernels.hip.h -- staging of alignment columns that are ALREADY in HBM (round 4).
//
// pc_add_alignment_file stages the caller's packed columns (tid, pos, alen, flags, nblk + the aligned runs of multi-run
// reads) with one threaded host pass -- validation, statistics, the 8-byte records, the run-stream records -- and the
// rest on the GPU.  The columns of a BAM file decoded on the GPU (bam_kernels.hip.h) are in HBM and were validated by
// the decoder: reading them back for that host pass and uploading the records again is a third of the time from file
// to counts.  Here the host pass itself is three kernels, two exclusive sums and a segmented maximum:
//
//   k_cols_runs     per record: its run count if its runs live in the run arrays (>= 2 runs), and if they also go to
//                   the run stream (aligned length <= 255, not wide)   -> exclusive sums: where a record's runs sit
//   k_cols_pack     per record: the 8-byte record {pos, length | flags | runs}, the run-stream records of its runs, its
//                   end; the statistics of the file -- span / length histograms in LDS per workgroup, flushed once
//   k_cols_bounds   per contig: its record range (the columns are sorted by contig) and its last record start
//   (segmented max) per contig: the furthest end of its reads
//
// What they produce is what the host pass produces (pc_add_alignment_file_wide in plastid_counts.hip, "ONE host pass"):
// the same records, the same statistics, hence the same staged file.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "plastid_counts.h"

namespace pcstage {

struct DevCols {   // device pointers
    const int32_t *tid, *pos;
    const uint16_t *alen;
    const uint8_t *flags, *nblk;
    const int32_t *blk_start, *blk_len;
    const uint32_t *wide_rec;   // ascending record indices of the wide records
    const uint2 *wide_val;      // their true {aligned length, run count}
    int64_t n_wide;
};

// statistics block (uint64 counters), in this order
constexpr int kSpanBins = 1026, kLenBins = 65536, kLen1Bins = 256;
constexpr int kAtSpan = 0, kAtGap = kAtSpan + kSpanBins, kAtWide = kAtGap + kSpanBins, kAtLen = kAtWide + kSpanBins, kAtLen1 = kAtLen + kLenBins,
              kAtMisc = kAtLen1 + kLen1Bins;   // misc: [0] Wr (max), [1] rmin (min), [2] rmax (max), [3] max_span (max)
constexpr int kStatWords = kAtMisc + 8;

__device__ __forceinline__ bool true_len(const DevCols &c, int64_t i, int64_t &L, int64_t &nb) {   // -> is the record wide?
    L = c.alen[i];
    nb = c.nblk[i];
    if (c.n_wide == 0 || L != 0xffff || nb != 0xff) return false;
    int64_t lo = 0, hi = c.n_wide;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)c.wide_rec[mid] < i) lo = mid + 1; else hi = mid;
    }
    if (lo < c.n_wide && (int64_t)c.wide_rec[lo] == i) { L = c.wide_val[lo].x; nb = c.wide_val[lo].y; return true; }
    return false;
}

__global__ __launch_bounds__(256) void k_cols_runs(DevCols c, int64_t n, uint32_t *__restrict__ in_arrays, uint32_t *__restrict__ in_stream) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i > n) return;
    uint32_t a = 0, s = 0;
    if (i < n) {
        int64_t L, nb;
        const bool wide = true_len(c, i, L, nb);
        a = nb >= 2 ? (uint32_t)nb : 0u;
        s = (nb >= 2 && L <= pc::kStreamMaxLen && !wide) ? (uint32_t)nb : 0u;
    }
    in_arrays[i] = a;   // (entry n: 0 -- the exclusive sums then end in the totals)
    in_stream[i] = s;
}

__global__ __launch_bounds__(256) void k_cols_pack(DevCols c, int64_t n, const uint32_t *__restrict__ cursor, const uint32_t *__restrict__ run_at,
                                                   uint2 *__restrict__ rec, uint2 *__restrict__ run_val, uint32_t *__restrict__ run_idx,
                                                   int32_t *__restrict__ ends, unsigned long long *__restrict__ stats) {
    __shared__ uint32_t h_span[kSpanBins], h_gap[kSpanBins], h_wide[kSpanBins], h_len[kSpanBins], h_len1[kLen1Bins];
    __shared__ uint32_t s_misc[4];
    for (int k = threadIdx.x; k < kSpanBins; k += 256) { h_span[k] = 0; h_gap[k] = 0; h_wide[k] = 0; h_len[k] = 0; }
    if (threadIdx.x < kLen1Bins) h_len1[threadIdx.x] = 0;
    if (threadIdx.x < 4) s_misc[threadIdx.x] = threadIdx.x == 1 ? 65536u : 0u;
    __syncthreads();
    uint32_t Wr = 0, rmin = 65536u, rmax = 0, max_span = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int64_t L, nb;
        const bool wide = true_len(c, i, L, nb);
        const int64_t p = c.pos[i];
        const uint32_t fl = pc::caller_flags(c.flags[i]);
        int64_t end;
        const uint32_t cur = cursor[i];
        if (nb >= 2) end = (int64_t)c.blk_start[cur + nb - 1] + c.blk_len[cur + nb - 1];
        else end = p + (L > 0 ? L : 1);
        const int64_t sp = end - p;
        ends[i] = (int32_t)end;
        const int sb = (int)(sp < 1025 ? sp : 1025);
        atomicAdd(&h_span[sb], 1u);
        if (L < kSpanBins) atomicAdd(&h_len[L], 1u);
        else atomicAdd(&stats[kAtLen + (L < 65535 ? L : 65535)], 1ull);
        max_span = sp > max_span ? (uint32_t)sp : max_span;
        uint32_t meta = (uint32_t)L | (fl << 16) | ((uint32_t)nb << 24);
        if (wide) meta = 0xffffu | ((fl | pc::kFlagWide) << 16) | (0xffu << 24);
        const bool in_runs = nb >= 2 && L <= pc::kStreamMaxLen && !wide;
        if (in_runs) {
            meta |= pc::kFlagRuns << 16;
            rmin = (uint32_t)L < rmin ? (uint32_t)L : rmin;
            rmax = (uint32_t)L > rmax ? (uint32_t)L : rmax;
            uint32_t cum = 0, at = run_at[i];
            for (int64_t k = 0; k < nb; ++k, ++at) {
                const uint32_t rs = (uint32_t)c.blk_start[cur + k], rl = (uint32_t)c.blk_len[cur + k];
                run_val[at] = make_uint2(rs, rl | (cum << 8) | ((uint32_t)L << 16) | (fl << 24));
                run_idx[at] = (uint32_t)i;
                Wr = rl > Wr ? rl : Wr;
                cum += rl;
            }
        } else if (wide) {
            atomicAdd(&h_wide[sb], 1u);
        } else {
            if (L > pc::kStreamMaxLen) atomicAdd(&h_gap[sb], 1u);
            else if (nb < 2) atomicAdd(&h_len1[L], 1u);
        }
        rec[i] = make_uint2((uint32_t)p, meta);
    }
    atomicMax(&s_misc[0], Wr);
    atomicMin(&s_misc[1], rmin);
    atomicMax(&s_misc[2], rmax);
    atomicMax(&s_misc[3], max_span);
    __syncthreads();
    for (int k = threadIdx.x; k < kSpanBins; k += 256) {
        if (h_span[k]) atomicAdd(&stats[kAtSpan + k], (unsigned long long)h_span[k]);
        if (h_gap[k]) atomicAdd(&stats[kAtGap + k], (unsigned long long)h_gap[k]);
        if (h_wide[k]) atomicAdd(&stats[kAtWide + k], (unsigned long long)h_wide[k]);
        if (h_len[k]) atomicAdd(&stats[kAtLen + k], (unsigned long long)h_len[k]);
    }
    if (threadIdx.x < kLen1Bins && h_len1[threadIdx.x]) atomicAdd(&stats[kAtLen1 + threadIdx.x], (unsigned long long)h_len1[threadIdx.x]);
    if (threadIdx.x == 0) {
        atomicMax(&stats[kAtMisc + 0], (unsigned long long)s_misc[0]);
        atomicMin(&stats[kAtMisc + 1], (unsigned long long)s_misc[1]);
        atomicMax(&stats[kAtMisc + 2], (unsigned long long)s_misc[2]);
        atomicMax(&stats[kAtMisc + 3], (unsigned long long)s_misc[3]);
    }
}

// bounds[t] = first record of contig t (t = 0 .. ntid; the columns are sorted by contig); last_pos[t] = start of its last record
__global__ __launch_bounds__(256) void k_cols_bounds(const int32_t *__restrict__ tid, const int32_t *__restrict__ pos, int64_t n, int ntid,
                                                     int64_t *__restrict__ bounds, int32_t *__restrict__ last_pos) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t > ntid) return;
    auto first_of = [&](int want) {
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (tid[mid] < want) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    const int64_t b = first_of(t);
    bounds[t] = b;
    if (t < ntid) {
        const int64_t en = first_of(t + 1);
        last_pos[t] = en > b ? pos[en - 1] : -1;
    }
}

} // namespace pcstage
