// pc_kernels.hip.h -- device code of the MI355X per-position read-counting engine.
//
// gfx950 only (wave64, 256 CUs in 8 XCDs, 160 KiB LDS/CU, HBM3E).  This path is
// integer/byte scatter-reduce work bound by HBM bandwidth: no MFMA anywhere.
//
// Data layout in HBM
//   rec[i]      uint2 {pos:int32, meta:uint32}, meta = L | flags<<16 | nblk<<24   (8 B / record,
//               BAM order; the only array the histogram kernel streams)
//   blk_off[i]  uint32, first run of record i (read only for the rare nblk>=2 records)
//   blk[j]      int2 {start,len}  aligned runs of the nblk>=2 records
//   hist        compact coverage over the *union* of queried intervals per strand
//               mode ("islands"): uint32 (point maps) or float64 (center);
//               rows x npos, row-major
//   out         the caller-visible int64/float64 vectors (every chain 5'->3')
//
// Kernels (one reference function each; reference = plastid/genomics/map_factories.pyx)
//   k_tile_ranges   fetch emulation: record range of every genome tile    (genome_array.py:800-809)
//   k_hist_point    FivePrime/ThreePrime/Variable/Stratified              (:308-367,:407-466,:585-650,:724-780)
//   k_long_point    same rules for the few long-span (spliced) reads
//   k_center        CenterMapFactory, ordered float64 replay              (:200-265)
//   k_gather        SegmentChain.get_counts layout + normalisation        (roitools.pyx:3259-3271,
//                                                                          genome_array.py:826-830)
//   k_mapped_reads  reads_out of the map functions for one segment
//   k_unmappable    records that make the reference emit its DataWarning
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pc {

constexpr int kWG = 256;          // 4 waves of 64
constexpr int kWave = 64;
constexpr uint32_t kFlagReverse = 0x01;
constexpr uint32_t kFlagLong = 0x40;      // engine-internal: span > W, handled by the long-read path
constexpr uint32_t kFlagExcluded = 0x80;
constexpr int kGatherChunk = 1024;

// strand modes of a query interval
//   0: '+'  keeps forward reads, forward index rule
//   1: '-'  keeps reverse reads, reverse index rule
//   2: '.'  keeps all reads,     forward index rule  (map_factories.pyx:345-346: only '-' flips)
//   3: all reads, reverse index rule (direct map-factory call on a '-' segment, no strand filter)
constexpr int kModes = 4;

struct FileView {
    const uint2 *rec;
    const uint32_t *blk_off;
    const int2 *blk;
    const int64_t *tid_bounds;      // ntid+1
    const uint32_t *long_idx;       // long-span records, record order
    const int32_t *long_tid;
    const int32_t *long_pmax;       // prefix max of ref_end within a tid
    const int64_t *long_tid_bounds; // ntid+1
    int64_t n;
    int64_t nlong;
};

struct MapParams {
    int kind;
    int param;
    int min_len, max_len;
    int rows;
    int filt_on, filt_min, filt_max;
    int table_len;
    const int32_t *fw;
    const int32_t *rc;
};

struct Tile {
    int32_t tid;
    int32_t win_start;
    uint32_t piece_begin;
    uint32_t piece_end;
    uint32_t mode_mask;
    uint32_t pad;
};

struct Piece {
    int64_t hist_off;
    int32_t start;
    int32_t len;
    int32_t mode;
    int32_t pad;
};

struct WorkItem {
    int64_t lo, hi;
    uint32_t tile;
    uint32_t file;
};

struct CenterChunk {
    int64_t hist_off;
    int32_t tid;
    int32_t start;
    int32_t len;
    int32_t mode;
};

struct GatherSeg {
    int64_t out_off;
    int64_t row_stride;
    int64_t hist_off; // hist index of position (start + clip_lo); -1: all zero
    int64_t len;
    int64_t clip_lo, clip_hi;
    int32_t step;
    int32_t pad;
};

struct GatherChunk {
    uint32_t seg;
    uint32_t chunk;
};

struct Unmappable {
    int32_t tid, pos, end, rev;
};

// ---------------------------------------------------------------- helpers
__device__ __forceinline__ int rec_len(uint32_t meta) { return (int)(meta & 0xffffu); }
__device__ __forceinline__ uint32_t rec_flags(uint32_t meta) { return (meta >> 16) & 0xffu; }
__device__ __forceinline__ int rec_nblk(uint32_t meta) { return (int)(meta >> 24); }

__device__ __forceinline__ bool size_ok(const MapParams &mp, int L) {
    // SizeFilterFactory.__call__, map_factories.pyx:837-839
    return !mp.filt_on || (L >= mp.filt_min && (L <= mp.filt_max || mp.filt_max == -1));
}

__device__ __forceinline__ bool strand_ok(int mode, bool rev) {
    // genome_array.py:812-815
    return mode == 0 ? !rev : (mode == 1 ? rev : true);
}

// first index in [lo,hi) whose pos >= key
__device__ __forceinline__ int64_t lower_bound_pos(const uint2 *rec, int64_t lo, int64_t hi, int64_t key) {
    while (lo < hi) {
        int64_t mid = lo + ((hi - lo) >> 1);
        if ((int64_t)(int32_t)rec[mid].x < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Index (counted from the left end of read.positions) the rule selects, or -1 when
// the read is not mapped.  `row` = output row (stratified).
__device__ __forceinline__ int map_kleft(const MapParams &mp, int L, bool rev_rule, int &row) {
    row = 0;
    switch (mp.kind) {
    case 0: // FivePrimeMapFactory.__call__ :343-355
        if (mp.param >= L) return -1;
        return rev_rule ? L - 1 - mp.param : mp.param;
    case 1: // ThreePrimeMapFactory.__call__ :442-454
        if (mp.param >= L) return -1;
        return rev_rule ? mp.param : L - 1 - mp.param;
    case 3: { // VariableFivePrimeMapFactory.__call__ :625-638
        if (L >= mp.table_len) return -1;
        return (rev_rule ? mp.rc : mp.fw)[L]; // -1 == _BAD_OFFSET
    }
    case 4: { // StratifiedVariableFivePrimeMapFactory.__call__ :765-778
        if (L < mp.min_len || L > mp.max_len || L < 1 || L >= mp.table_len) return -1;
        int off = (rev_rule ? mp.rc : mp.fw)[L];
        row = L - mp.min_len;
        return off < 0 ? L - 1 : off; // no _BAD_OFFSET check: read_positions[-1]
    }
    default:
        return -1;
    }
}

// read.positions[k] for a record with aligned runs
__device__ __forceinline__ int32_t walk_runs(const FileView &fv, int64_t i, int nblk, int k) {
    const int2 *b = fv.blk + fv.blk_off[i];
    int32_t p = 0;
    for (int j = 0; j < nblk; ++j) {
        int2 r = b[j];
        if (k < r.y) { p = r.x + k; break; }
        k -= r.y;
    }
    return p;
}

__device__ __forceinline__ int32_t rec_end(const FileView &fv, int64_t i, int32_t pos, uint32_t meta) {
    // htslib bam_endpos
    int nb = rec_nblk(meta);
    if (nb >= 2) {
        int2 r = fv.blk[fv.blk_off[i] + nb - 1];
        return r.x + r.y;
    }
    int L = rec_len(meta);
    return pos + (L > 0 ? L : 1);
}

// ---------------------------------------------------------------- k_tile_ranges
// One thread per (tile, file): the record range a tile has to scan, cut into work
// items of at most `R` records (load balance for pile-ups).
__global__ __launch_bounds__(kWG) void k_tile_ranges(const Tile *__restrict__ tiles, int ntiles,
                                                     const FileView *__restrict__ files, int nfiles,
                                                     int G, int W, int64_t R, WorkItem *work,
                                                     uint32_t *nwork, uint32_t *tile_items,
                                                     uint32_t work_cap) {
    int64_t idx = (int64_t)blockIdx.x * kWG + threadIdx.x;
    if (idx >= (int64_t)ntiles * nfiles) return;
    int t = (int)(idx / nfiles), f = (int)(idx % nfiles);
    Tile tl = tiles[t];
    const FileView &fv = files[f];
    int64_t b0 = fv.tid_bounds[tl.tid], b1 = fv.tid_bounds[tl.tid + 1];
    int64_t lo = lower_bound_pos(fv.rec, b0, b1, (int64_t)tl.win_start - W + 1);
    int64_t hi = lower_bound_pos(fv.rec, lo, b1, (int64_t)tl.win_start + G);
    int64_t n = hi - lo;
    if (n <= 0) return;
    uint32_t items = (uint32_t)((n + R - 1) / R);
    uint32_t base = atomicAdd(nwork, items);
    atomicAdd(&tile_items[t], items);
    for (uint32_t k = 0; k < items; ++k) {
        if (base + k >= work_cap) break; // cannot happen (capacity is an upper bound); defensive
        WorkItem w;
        w.lo = lo + (int64_t)k * R;
        w.hi = (w.lo + R < hi) ? w.lo + R : hi;
        w.tile = (uint32_t)t;
        w.file = (uint32_t)f;
        work[base + k] = w;
    }
}

// ---------------------------------------------------------------- k_hist_point
// One workgroup per work item.  Streams its records once (coalesced 8-byte
// loads), bins every read's mapped position with LDS atomics into a window of G
// genome positions per strand mode, then writes the island pieces of the window
// to the compact histogram (plain coalesced stores when the tile has a single
// work item, global atomics otherwise).
__device__ __forceinline__ void hist_one(const FileView &fv, const MapParams &mp, int64_t i, uint2 r,
                                         const int *slot, int32_t win_start, int G, uint32_t *bins) {
    const uint32_t meta = r.y;
    const uint32_t fl = rec_flags(meta);
    if (fl & (kFlagExcluded | kFlagLong)) return;
    const int L = rec_len(meta);
    if (!size_ok(mp, L)) return;
    const bool rev = fl & kFlagReverse;
    const int nb = rec_nblk(meta);
    const int32_t pos = (int32_t)r.x;
    int row_f, row_r;
    const int kf = map_kleft(mp, L, false, row_f);
    const int kr = map_kleft(mp, L, true, row_r);
    int32_t pf = 0, pr = 0;
    if (nb >= 2) {
        if (kf >= 0) pf = walk_runs(fv, i, nb, kf);
        if (kr >= 0) pr = walk_runs(fv, i, nb, kr);
    } else {
        pf = pos + kf;
        pr = pos + kr;
    }
#pragma unroll
    for (int m = 0; m < kModes; ++m) {
        if (slot[m] < 0 || !strand_ok(m, rev)) continue;
        const bool rr = (m == 1 || m == 3);
        const int k = rr ? kr : kf;
        if (k < 0) continue;
        const uint32_t d = (uint32_t)((rr ? pr : pf) - win_start);
        if (d < (uint32_t)G) atomicAdd(&bins[(size_t)(slot[m] * mp.rows + (rr ? row_r : row_f)) * G + d], 1u);
    }
}

__global__ __launch_bounds__(kWG) void k_hist_point(const Tile *__restrict__ tiles,
                                                    const Piece *__restrict__ pieces,
                                                    const FileView *__restrict__ files,
                                                    const WorkItem *__restrict__ work,
                                                    const uint32_t *__restrict__ nwork,
                                                    const uint32_t *__restrict__ tile_items, MapParams mp,
                                                    int G, uint32_t *hist, int64_t hist_row_stride) {
    extern __shared__ __attribute__((aligned(16))) uint32_t bins[];
    if (blockIdx.x >= *nwork) return;
    const WorkItem w = work[blockIdx.x];
    const Tile tl = tiles[w.tile];
    const FileView &fv = files[w.file];
    int slot[kModes];
    int nslots = 0;
#pragma unroll
    for (int m = 0; m < kModes; ++m) slot[m] = ((tl.mode_mask >> m) & 1u) ? nslots++ : -1;
    const int nbins = nslots * mp.rows * G;
    for (int i = threadIdx.x; i < nbins; i += kWG) bins[i] = 0;
    __syncthreads();

    constexpr int U = 4;
    for (int64_t base = w.lo; base < w.hi; base += (int64_t)kWG * U) {
        uint2 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int64_t i = base + u * kWG + threadIdx.x;
            r[u] = (i < w.hi) ? fv.rec[i] : make_uint2(0u, kFlagExcluded << 16);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) hist_one(fv, mp, base + u * kWG + threadIdx.x, r[u], slot, tl.win_start, G, bins);
    }
    __syncthreads();

    const bool single = tile_items[w.tile] == 1u;
    for (uint32_t pi = tl.piece_begin; pi < tl.piece_end; ++pi) {
        const Piece pc_ = pieces[pi];
        const int s = slot[pc_.mode];
        const int rel = pc_.start - tl.win_start;
        for (int r = 0; r < mp.rows; ++r) {
            const uint32_t *src = bins + (size_t)(s * mp.rows + r) * G + rel;
            uint32_t *dst = hist + (size_t)r * hist_row_stride + pc_.hist_off;
            if (single) {
                for (int i = threadIdx.x; i < pc_.len; i += kWG) dst[i] = src[i];
            } else {
                for (int i = threadIdx.x; i < pc_.len; i += kWG) {
                    uint32_t v = src[i];
                    if (v) atomicAdd(&dst[i], v);
                }
            }
        }
    }
}

// ---------------------------------------------------------------- k_long_point
// Long-span (spliced) reads are skipped by the window scan; one thread per such
// read computes its mapped position, finds the island piece holding it and adds
// with a global atomic.
__device__ __forceinline__ int64_t find_tile(const Tile *tiles, int ntiles, int32_t tid, int32_t win_start) {
    int64_t lo = 0, hi = ntiles;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        Tile t = tiles[mid];
        bool less = t.tid < tid || (t.tid == tid && t.win_start < win_start);
        if (less) lo = mid + 1; else hi = mid;
    }
    if (lo < ntiles && tiles[lo].tid == tid && tiles[lo].win_start == win_start) return lo;
    return -1;
}

__global__ __launch_bounds__(kWG) void k_long_point(const Tile *__restrict__ tiles, int ntiles,
                                                    const Piece *__restrict__ pieces, FileView fv,
                                                    MapParams mp, int G, uint32_t plan_modes,
                                                    uint32_t *hist, int64_t hist_row_stride) {
    int64_t j = (int64_t)blockIdx.x * kWG + threadIdx.x;
    if (j >= fv.nlong) return;
    const int64_t i = fv.long_idx[j];
    const uint2 r = fv.rec[i];
    const uint32_t meta = r.y;
    const uint32_t fl = rec_flags(meta);
    if (fl & kFlagExcluded) return;
    const int L = rec_len(meta);
    if (!size_ok(mp, L)) return;
    const bool rev = fl & kFlagReverse;
    const int nb = rec_nblk(meta);
    const int32_t tid = fv.long_tid[j];
    for (int m = 0; m < kModes; ++m) {
        if (!((plan_modes >> m) & 1u) || !strand_ok(m, rev)) continue;
        int row;
        const int k = map_kleft(mp, L, m == 1 || m == 3, row);
        if (k < 0) continue;
        const int32_t p = nb >= 2 ? walk_runs(fv, i, nb, k) : (int32_t)r.x + k;
        const int64_t t = find_tile(tiles, ntiles, tid, (int32_t)(((int64_t)p / G) * G));
        if (t < 0) continue;
        const Tile tl = tiles[t];
        if (!((tl.mode_mask >> m) & 1u)) continue;
        for (uint32_t pi = tl.piece_begin; pi < tl.piece_end; ++pi) {
            const Piece pc_ = pieces[pi];
            if (pc_.mode == m && p >= pc_.start && p < pc_.start + pc_.len) {
                atomicAdd(&hist[(size_t)row * hist_row_stride + pc_.hist_off + (p - pc_.start)], 1u);
                break;
            }
        }
    }
}

// ---------------------------------------------------------------- k_center
// CenterMapFactory: count[p] is the left-to-right float64 sum, in read order, of
// 1/(L-2*nibble) over the reads whose trimmed positions contain p.  The order is
// part of the contract (the reference's own test demands exact equality), so
// there are no atomics: one lane owns one output position and replays, in record
// order, every read that can cover it.  One wave per 64 positions; the candidate
// loop is wave-uniform (scalar loads), the coverage test is per lane.
__device__ __forceinline__ void center_one(const FileView &fv, const MapParams &mp, int64_t i, int mode,
                                           const double *__restrict__ inv, int32_t p, double &acc) {
    const uint2 r = fv.rec[i];
    const uint32_t meta = r.y;
    const uint32_t fl = rec_flags(meta);
    if (fl & kFlagExcluded) return;
    if (!strand_ok(mode, fl & kFlagReverse)) return;
    const int L = rec_len(meta);
    if (!size_ok(mp, L)) return;
    const int nib = mp.param;
    const int m = L - 2 * nib;              // map_length, :245
    if (m <= 0) return;                     // :246-249
    const double val = inv[m];              // 1.0 / map_length, :250 (host-computed IEEE quotient)
    const int nb = rec_nblk(meta);
    bool hit;
    if (nb < 2) {
        const int32_t s = (int32_t)r.x + nib;
        hit = p >= s && p < s + m;
    } else {
        hit = false;
        const int2 *b = fv.blk + fv.blk_off[i];
        int cum = 0;
        for (int j = 0; j < nb; ++j) {
            const int2 run = b[j];
            const int idx = cum + (p - run.x);
            hit |= (p >= run.x) && (p < run.x + run.y) && (idx >= nib) && (idx < L - nib);
            cum += run.y;
        }
    }
    if (hit) acc += val;                    // :254, one IEEE add per covering read, in order
}

__global__ __launch_bounds__(kWG) void k_center(const CenterChunk *__restrict__ chunks, int64_t nchunks,
                                                const FileView *__restrict__ files, int nfiles,
                                                MapParams mp, int W, const double *__restrict__ inv,
                                                double *hist) {
    const int64_t c = __builtin_amdgcn_readfirstlane((int)(((int64_t)blockIdx.x * kWG + threadIdx.x) >> 6));
    if (c >= nchunks) return;
    const int lane = threadIdx.x & 63;
    const CenterChunk ck = chunks[c];
    const int32_t p = ck.start + lane;
    double acc = 0.0;
    for (int f = 0; f < nfiles; ++f) { // file-major, genome_array.py:800-809
        const FileView &fv = files[f];
        const int64_t b0 = fv.tid_bounds[ck.tid], b1 = fv.tid_bounds[ck.tid + 1];
        const int64_t near_key = (int64_t)ck.start - W + 1;
        if (fv.nlong) {
            // long-span reads that start before the near window but may reach into it
            const int64_t l0 = fv.long_tid_bounds[ck.tid], l1 = fv.long_tid_bounds[ck.tid + 1];
            int64_t lo = l0, hi = l1;
            while (lo < hi) { // first long read with pos >= near_key
                int64_t mid = lo + ((hi - lo) >> 1);
                if ((int64_t)(int32_t)fv.rec[fv.long_idx[mid]].x < near_key) lo = mid + 1; else hi = mid;
            }
            const int64_t jhi = lo;
            lo = l0; hi = jhi;
            while (lo < hi) { // first long read whose running max end reaches past the chunk start
                int64_t mid = lo + ((hi - lo) >> 1);
                if (fv.long_pmax[mid] <= ck.start) lo = mid + 1; else hi = mid;
            }
            for (int64_t j = lo; j < jhi; ++j) center_one(fv, mp, fv.long_idx[j], ck.mode, inv, p, acc);
        }
        const int64_t lo = lower_bound_pos(fv.rec, b0, b1, near_key);
        const int64_t hi = lower_bound_pos(fv.rec, lo, b1, (int64_t)ck.start + ck.len);
        for (int64_t i = lo; i < hi; ++i) center_one(fv, mp, i, ck.mode, inv, p, acc);
    }
    if (lane < ck.len) hist[ck.hist_off + lane] = acc;
}

// ---------------------------------------------------------------- k_gather
// Lays the per-segment slices out the way SegmentChain.get_counts does: chain
// offset, 5'->3' reversal for '-' chains, int64 or float64, optional
// reads-per-million normalisation (count / sum * 1e6, in that order).
template <typename HistT, typename OutT, bool NORM>
__global__ __launch_bounds__(kWG) void k_gather(const GatherSeg *__restrict__ segs,
                                                const GatherChunk *__restrict__ chunks,
                                                const HistT *__restrict__ hist, int64_t hist_row_stride,
                                                int rows, double norm_sum, OutT *out) {
    const GatherChunk gc = chunks[blockIdx.x];
    const GatherSeg sg = segs[gc.seg];
    const int64_t base = (int64_t)gc.chunk * kGatherChunk;
    const int64_t n = (sg.len - base < kGatherChunk) ? sg.len - base : kGatherChunk;
    for (int r = 0; r < rows; ++r) {
        const HistT *src = hist + (size_t)r * hist_row_stride + sg.hist_off - sg.clip_lo;
        OutT *dst = out + sg.out_off + (int64_t)r * sg.row_stride;
        for (int64_t i = threadIdx.x; i < n; i += kWG) {
            const int64_t idx = base + i;
            HistT v = 0;
            if (sg.hist_off >= 0 && idx >= sg.clip_lo && idx < sg.clip_hi) v = src[idx];
            OutT o;
            if (NORM) o = (OutT)((double)v / norm_sum * 1e6);
            else o = (OutT)v;
            dst[(int64_t)sg.step * idx] = o;
        }
    }
}

// ---------------------------------------------------------------- totals
__global__ __launch_bounds__(kWG) void k_total_i64(const int64_t *__restrict__ x, int64_t n, int64_t *total) {
    int64_t s = 0;
    for (int64_t i = (int64_t)blockIdx.x * kWG + threadIdx.x; i < n; i += (int64_t)gridDim.x * kWG) s += x[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd((unsigned long long *)total, (unsigned long long)s);
}

// fixed-order float64 sum: each block reduces a fixed slice with a fixed tree,
// block partials are then summed by one thread in block order.
__global__ __launch_bounds__(kWG) void k_total_f64_partial(const double *__restrict__ x, int64_t n, double *partial) {
    __shared__ double sm[kWG];
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t b = (int64_t)blockIdx.x * per;
    const int64_t e = (b + per < n) ? b + per : n;
    double s = 0.0;
    for (int64_t i = b + threadIdx.x; i < e; i += kWG) s += x[i];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = kWG / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}

__global__ void k_total_f64_final(const double *__restrict__ partial, int nb, double *total) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < nb; ++i) s += partial[i];
        *total = s;
    }
}

// ---------------------------------------------------------------- k_mapped_reads
// reads_out of the map functions for ONE segment (genome_array.py:800-823).
__global__ __launch_bounds__(kWG) void k_mapped_reads(FileView fv, MapParams mp, int64_t rec_lo, int64_t rec_hi,
                                                      int64_t start, int64_t end, int mode, bool strand_filter,
                                                      uint8_t *mask) {
    int64_t i = rec_lo + (int64_t)blockIdx.x * kWG + threadIdx.x;
    if (i >= rec_hi) return;
    const uint2 r = fv.rec[i];
    const uint32_t meta = r.y;
    const uint32_t fl = rec_flags(meta);
    const int32_t pos = (int32_t)r.x;
    const int L = rec_len(meta);
    const int nb = rec_nblk(meta);
    const bool rev = fl & kFlagReverse;
    uint8_t out = 0;
    const bool fetched = (int64_t)pos < end && (int64_t)rec_end(fv, i, pos, meta) > start;
    if (fetched && !(fl & kFlagExcluded) && (!strand_filter || strand_ok(mode, rev)) && size_ok(mp, L)) {
        if (mp.kind == 2) {
            out = (L - 2 * mp.param) > 0; // CenterMapFactory :249-256: appended even if nothing landed
        } else {
            int row;
            const int k = map_kleft(mp, L, mode == 1 || mode == 3, row);
            if (k >= 0) {
                const int64_t p = nb >= 2 ? walk_runs(fv, i, nb, k) : pos + k;
                out = p >= start && p < end;
            }
        }
    }
    mask[i - rec_lo] = out;
}

// ---------------------------------------------------------------- k_unmappable
// Records for which the reference sets its warning flag (:246-248, :351-353,
// :450-452, :633-636); compacted for the host-side per-segment overlap test.
__global__ __launch_bounds__(kWG) void k_unmappable(FileView fv, MapParams mp, int ntid, Unmappable *list,
                                                    uint32_t cap, uint32_t *count) {
    int64_t i = (int64_t)blockIdx.x * kWG + threadIdx.x;
    if (i >= fv.n) return;
    const uint2 r = fv.rec[i];
    const uint32_t meta = r.y;
    const uint32_t fl = rec_flags(meta);
    const int L = rec_len(meta);
    if ((fl & kFlagExcluded) || !size_ok(mp, L)) return;
    bool bad;
    switch (mp.kind) {
    case 0: case 1: bad = mp.param >= L; break;
    case 2: bad = L - 2 * mp.param < 0; break;
    case 3: bad = L >= mp.table_len || mp.fw[L] < 0; break;
    default: bad = false;
    }
    if (!bad) return;
    uint32_t slot = atomicAdd(count, 1u);
    if (slot >= cap) return;
    // tid by binary search over the per-tid record bounds
    int lo = 0, hi = ntid;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (fv.tid_bounds[mid + 1] <= i) lo = mid + 1; else hi = mid;
    }
    Unmappable u;
    u.tid = lo;
    u.pos = (int32_t)r.x;
    u.end = rec_end(fv, i, (int32_t)r.x, meta);
    u.rev = (fl & kFlagReverse) ? 1 : 0;
    list[slot] = u;
}

} // namespace pc
