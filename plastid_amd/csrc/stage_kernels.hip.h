// stage_kernels.hip.h -- staging of alignment columns that are in HBM (round 4: the columns of a BAM file decoded on
// the GPU; round 5: the caller's own host arrays too).
//
// Until round 5 pc_add_alignment_file staged the caller's packed columns (tid, pos, alen, flags, nblk + the aligned runs
// of multi-run reads) with one threaded host pass -- validation, statistics, the 8-byte records, the run-stream records
// -- that no number of host threads brought near the rate of the PCIe link.  Now the host only moves bytes: the columns
// cross PCIe as they are (through a ring of page-locked pieces, plastid_counts.hip "TransferRing"; the contig column does
// not travel at all: sorted, it is ntid + 1 record bounds, found by a streaming comparison on the host while the other
// columns are in flight), and everything the host pass did is done here, the validation included (k_cols_pack<true>).
// The columns of a BAM file decoded on the GPU (bam_kernels.hip.h) are in HBM to begin with and were validated by the
// decoder (k_cols_pack<false>).  The kernels, with three exclusive sums between them:
//
//   k_cols_runs     per record: its run count if its runs live in the run arrays (>= 2 runs), and if they also go to
//                   the run stream (aligned length <= 255, not wide)   -> exclusive sums: where a record's runs sit
//   k_cols_pack     per record: the 8-byte record {pos, length | flags | runs}, the run-stream records of its runs; the
//                   statistics of the file -- span / length histograms in LDS per workgroup, flushed once -- and the
//                   furthest end of the reads of every contig
//   k_cols_bounds   (decoded columns) per contig: its record range and its last record start
//   k_classify      per record: the long-span class (the halo is a quantile of the spans of the whole file), the 4-byte
//                   stream word; per workgroup the members of the three side lists
//   k_side_select   the record indices of the side lists, in record order
//
// Errors of caller-owned columns are reported as the host pass reported them: the defect of the lowest record index,
// and of the first check that fails for that record (an atomic minimum over `record << 8 | check`).
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

// checks of k_cols_pack<true>, in the order a record is examined (the contig column is examined on the host: kBadTid)
enum { kBadTid = 1, kBadNegPos, kBadOrder, kBadRuns, kBadFirstRun, kBadRunSum, kBadLenRuns, kBadEnd };

// `totals` (caller-owned columns): the two sums in 64 bits -- run counts a caller made up can wrap the 32-bit exclusive
// sums below into agreement with the length of the run arrays; whoever trusts those sums checks these first
__global__ __launch_bounds__(256) void k_cols_runs(DevCols c, int64_t n, uint32_t *__restrict__ in_arrays, uint32_t *__restrict__ in_stream,
                                                   unsigned long long *__restrict__ totals) {
    unsigned long long wa = 0, ws = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i <= n; i += (int64_t)gridDim.x * 256) {
        uint32_t a = 0, s = 0;
        if (i < n) {
            int64_t L, nb;
            const bool wide = true_len(c, i, L, nb);
            a = nb >= 2 ? (uint32_t)nb : 0u;
            s = (nb >= 2 && L <= pc::kStreamMaxLen && !wide) ? (uint32_t)nb : 0u;
        }
        in_arrays[i] = a;   // (entry n: 0 -- the exclusive sums then end in the totals)
        in_stream[i] = s;
        wa += a;
        ws += s;
    }
    if (!totals) return;
    // (a few thousand workgroups, one pair of atomics each: one per 256 records was 4 M atomics on two addresses, 45 ms)
    __shared__ unsigned long long t_a, t_s;
    if (threadIdx.x == 0) { t_a = 0; t_s = 0; }
    __syncthreads();
    for (int d = 32; d >= 1; d >>= 1) { wa += __shfl_down(wa, d); ws += __shfl_down(ws, d); }
    if ((threadIdx.x & 63) == 0 && (wa | ws)) { atomicAdd(&t_a, wa); atomicAdd(&t_s, ws); }
    __syncthreads();
    if (threadIdx.x == 0 && (t_a | t_s)) { atomicAdd(&totals[0], t_a); atomicAdd(&totals[1], t_s); }
}

// CHECK: the columns are the caller's -- every record is validated as the host pass of earlier rounds validated it (c.tid
// is not read: the contig column stayed on the host, `bounds` is what it came to); a record with a defect is reported in
// `err` and left out.
// A workgroup packs one contiguous stretch of the records, contig by contig (`bounds`), and leaves the furthest end of
// each contig's reads in tid_end with one atomic per (workgroup, contig) -- the per-contig maxima were a segmented
// reduction over an array of all ends before: 13 ms for 500 M records in 25 contigs, one workgroup per contig.
template <bool CHECK>
__global__ __launch_bounds__(256) void k_cols_pack(DevCols c, int64_t n, const uint32_t *__restrict__ cursor, const uint32_t *__restrict__ run_at,
                                                   uint2 *__restrict__ rec, uint2 *__restrict__ run_val, uint32_t *__restrict__ run_idx,
                                                   int32_t *__restrict__ tid_end, unsigned long long *__restrict__ stats,
                                                   const int64_t *__restrict__ bounds, int ntid, unsigned long long *__restrict__ err) {
    __shared__ uint32_t h_span[kSpanBins], h_gap[kSpanBins], h_wide[kSpanBins], h_len[kSpanBins], h_len1[kLen1Bins];
    __shared__ uint32_t s_misc[4];
    __shared__ int32_t s_end[4];
    for (int k = threadIdx.x; k < kSpanBins; k += 256) { h_span[k] = 0; h_gap[k] = 0; h_wide[k] = 0; h_len[k] = 0; }
    if (threadIdx.x < kLen1Bins) h_len1[threadIdx.x] = 0;
    if (threadIdx.x < 4) s_misc[threadIdx.x] = threadIdx.x == 1 ? 65536u : 0u;
    __syncthreads();
    uint32_t Wr = 0, rmin = 65536u, rmax = 0, max_span = 0;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t b0 = (int64_t)blockIdx.x * per < n ? (int64_t)blockIdx.x * per : n, b1 = b0 + per < n ? b0 + per : n;
    int t = 0;   // the contig of record b0: the last one that starts at or before it
    {
        int lo = 0, hi = ntid;   // first t in [0, ntid] with bounds[t] > b0
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (bounds[mid] <= b0) lo = mid + 1; else hi = mid;
        }
        t = lo > 0 ? lo - 1 : 0;
    }
    for (int64_t s0 = b0; s0 < b1;) {
        while (t + 1 < ntid && bounds[t + 1] <= s0) ++t;
        const int64_t s1 = bounds[t + 1] < b1 ? bounds[t + 1] : b1;   // (bounds[ntid] = n >= b1)
        int32_t emax = 0;
        for (int64_t i = s0 + threadIdx.x; i < s1; i += 256) {
            int64_t L, nb;
            const bool wide = true_len(c, i, L, nb);
            const int64_t p = c.pos[i];
            const uint32_t fl = pc::caller_flags(c.flags[i]);
            int64_t end = 0;
            const uint32_t cur = cursor[i];
            if (CHECK) {
                int bad = 0;
                if (p < 0) bad = kBadNegPos;
                else if (i > bounds[t] && p < (int64_t)c.pos[i - 1]) bad = kBadOrder;   // (positions start over with a contig)
                else if (nb >= 2) {
                    int64_t sum = 0, prev_end = -1;
                    for (int64_t k = 0; k < nb && !bad; ++k) {
                        const int64_t r0 = c.blk_start[cur + k], ln = c.blk_len[cur + k];
                        if (ln <= 0 || (k > 0 && r0 <= prev_end)) bad = kBadRuns;
                        else if (k == 0 && r0 != p) bad = kBadFirstRun;
                        sum += ln;
                        prev_end = r0 + ln;
                    }
                    if (!bad && sum != L) bad = kBadRunSum;
                    end = prev_end;
                } else {
                    if ((nb == 0) != (L == 0)) bad = kBadLenRuns;
                    end = p + (L > 0 ? L : 1);
                }
                if (!bad && end > 0x7fffffffLL) bad = kBadEnd;
                if (bad) {
                    atomicMin(err, ((unsigned long long)i << 8) | (unsigned long long)bad);
                    continue;
                }
            } else if (nb >= 2) end = (int64_t)c.blk_start[cur + nb - 1] + c.blk_len[cur + nb - 1];
            else end = p + (L > 0 ? L : 1);
            const int64_t sp = end - p;
            emax = (int32_t)end > emax ? (int32_t)end : emax;
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
        for (int d = 32; d >= 1; d >>= 1) { const int32_t o = __shfl_down(emax, d); emax = o > emax ? o : emax; }
        if ((threadIdx.x & 63) == 0) s_end[threadIdx.x >> 6] = emax;
        __syncthreads();
        if (threadIdx.x == 0) {
            int32_t m = s_end[0];
            for (int w = 1; w < 4; ++w) m = s_end[w] > m ? s_end[w] : m;
            if (m > 0) atomicMax(&tid_end[t], m);
        }
        __syncthreads();
        s0 = s1;
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

// ---- the class of every record that depends on the statistics of the whole file -- kFlagLong when its span (pos .. end of
// the last aligned run) is beyond the window halo `wcap`, always for a wide record -- written into the header, and the
// 4-byte stream word that follows from the header.  The same pass counts, per workgroup, the members of the three side
// lists (long-span, gapped, long-span outside the run stream): `counts[c * gridDim.x + workgroup]`, whose exclusive sum
// tells k_side_select where every workgroup's members go.
__device__ __forceinline__ uint32_t side_classes(uint32_t meta) {   // bit 0 long-span list, 1 gapped-record list, 2 long-span outside the run stream
    const uint32_t fl = meta >> 16;
    if (fl & pc::kFlagLong) return (fl & pc::kFlagRuns) ? 1u : 5u;
    return (!(fl & pc::kFlagRuns) && ((meta >> 24) >= 2u || (meta & 0xffffu) > (uint32_t)pc::kStreamMaxLen)) ? 2u : 0u;
}

__global__ __launch_bounds__(256) void k_classify(uint2 *rec, int64_t n, const uint32_t *__restrict__ blk_off, const int2 *__restrict__ blk,
                                                  int wcap, uint32_t *__restrict__ stream, uint32_t *__restrict__ counts) {
    __shared__ uint32_t s_cnt[3];
    if (threadIdx.x < 3) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t cls = 0;
    if (i < n) {
        uint2 r = rec[i];
        const uint32_t L = r.y & 0xffffu, fl = (r.y >> 16) & 0xffu, nb = r.y >> 24;
        bool far = (fl & pc::kFlagWide) != 0u;
        if (!far) {
            int64_t span = L > 0u ? (int64_t)L : 1;
            if (nb >= 2u) {
                const int2 last = blk[blk_off[i] + nb - 1u];
                span = (int64_t)last.x + last.y - (int64_t)(int32_t)r.x;
            }
            far = span > (int64_t)wcap;
        }
        if (far) {
            r.y |= pc::kFlagLong << 16;
            rec[i].y = r.y;
        }
        stream[i] = pc::stream_word(r.x, r.y);
        cls = side_classes(r.y);
    }
    for (int k = 0; k < 3; ++k) {
        const unsigned long long m = __ballot((cls >> k) & 1u);
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(&s_cnt[k], (uint32_t)__popcll(m));
    }
    __syncthreads();
    if (threadIdx.x < 3) counts[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s_cnt[threadIdx.x];
}

// the record indices of the three lists, in record order: `offs` = exclusive sum of k_classify's counts (3 * nwg + 1 entries)
__global__ __launch_bounds__(256) void k_side_select(const uint2 *__restrict__ rec, int64_t n, const uint32_t *__restrict__ offs, uint32_t nwg,
                                                     uint32_t *__restrict__ long_idx, uint32_t *__restrict__ gap_idx, uint32_t *__restrict__ xlong_idx) {
    const uint32_t b = blockIdx.x;
    const uint32_t o0 = offs[b], o1 = offs[nwg + b], o2 = offs[2 * nwg + b];
    if (offs[b + 1] == o0 && offs[nwg + b + 1] == o1 && offs[2 * nwg + b + 1] == o2) return;   // (most workgroups of an ungapped file)
    __shared__ uint32_t s_w[3][4];
    const int64_t i = (int64_t)b * 256 + threadIdx.x;
    const uint32_t cls = i < n ? side_classes(rec[i].y) : 0u;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long m[3];
    for (int k = 0; k < 3; ++k) {
        m[k] = __ballot((cls >> k) & 1u);
        if (lane == 0) s_w[k][w] = (uint32_t)__popcll(m[k]);
    }
    __syncthreads();
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    uint32_t *const out[3] = {long_idx, gap_idx, xlong_idx};
    const uint32_t at[3] = {o0, o1 - offs[nwg], o2 - offs[2 * nwg]};
    for (int k = 0; k < 3; ++k)
        if ((cls >> k) & 1u) {
            uint32_t before = 0;
            for (int v = 0; v < w; ++v) before += s_w[k][v];
            out[k][at[k] + before + (uint32_t)__popcll(m[k] & below)] = (uint32_t)i;
        }
}

} // namespace pcstage
