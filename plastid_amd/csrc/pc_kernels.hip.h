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
    const uint4 *gap_rec;           // short-span gapped records {pos, meta, blk_off, rec_idx}, record order
    const int64_t *gap_tid_bounds;  // ntid+1
    int64_t n;
    int64_t nlong;
    int64_t ngap;
};

// Pointers that come out of a FileView are loaded from memory, so the compiler only knows
// them as generic ("flat") pointers: flat loads count on both vmcnt and lgkmcnt and cannot be
// scalarised.  GFile re-types them as global (address space 1) once per kernel.
#define PC_GLOBAL __attribute__((address_space(1)))
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

struct GFile {
    const u32x2 PC_GLOBAL *rec;
    const u32x4 PC_GLOBAL *rec4;
    const uint32_t PC_GLOBAL *blk_off;
    const i32x2 PC_GLOBAL *blk;
    const int64_t PC_GLOBAL *tid_bounds;
    const uint32_t PC_GLOBAL *long_idx;
    const int32_t PC_GLOBAL *long_tid;
    const int32_t PC_GLOBAL *long_pmax;
    const int64_t PC_GLOBAL *long_tid_bounds;
    const u32x4 PC_GLOBAL *gap_rec;
    const int64_t PC_GLOBAL *gap_tid_bounds;
    int64_t n;
    int64_t nlong;
    int64_t ngap;
};

__device__ __forceinline__ GFile gfile(const FileView &v) {
    GFile g;
    g.rec = (const u32x2 PC_GLOBAL *)v.rec;
    g.rec4 = (const u32x4 PC_GLOBAL *)v.rec;
    g.blk_off = (const uint32_t PC_GLOBAL *)v.blk_off;
    g.blk = (const i32x2 PC_GLOBAL *)v.blk;
    g.tid_bounds = (const int64_t PC_GLOBAL *)v.tid_bounds;
    g.long_idx = (const uint32_t PC_GLOBAL *)v.long_idx;
    g.long_tid = (const int32_t PC_GLOBAL *)v.long_tid;
    g.long_pmax = (const int32_t PC_GLOBAL *)v.long_pmax;
    g.long_tid_bounds = (const int64_t PC_GLOBAL *)v.long_tid_bounds;
    g.gap_rec = (const u32x4 PC_GLOBAL *)v.gap_rec;
    g.gap_tid_bounds = (const int64_t PC_GLOBAL *)v.gap_tid_bounds;
    g.n = v.n;
    g.nlong = v.nlong;
    g.ngap = v.ngap;
    return g;
}

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
    int64_t lo, hi;   // record range of the packed stream
    int64_t glo, ghi; // range of the gapped-record list (first work item of a tile only)
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
__device__ __forceinline__ int64_t lower_bound_pos(const u32x2 PC_GLOBAL *rec, int64_t lo, int64_t hi, int64_t key) {
    while (lo < hi) {
        int64_t mid = lo + ((hi - lo) >> 1);
        if ((int64_t)(int32_t)rec[mid].x < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Index (counted from the left end of read.positions) the rule selects, or -1 when
// the read is not mapped.  `row` = output row (stratified).
template <int KIND>
__device__ __forceinline__ int map_kleft(const MapParams &mp, int L, bool rev_rule, int &row) {
    row = 0;
    if (KIND == 0) { // FivePrimeMapFactory.__call__ :343-355
        if (mp.param >= L) return -1;
        return rev_rule ? L - 1 - mp.param : mp.param;
    } else if (KIND == 1) { // ThreePrimeMapFactory.__call__ :442-454
        if (mp.param >= L) return -1;
        return rev_rule ? mp.param : L - 1 - mp.param;
    } else if (KIND == 3) { // VariableFivePrimeMapFactory.__call__ :625-638
        if (L >= mp.table_len) return -1;
        const int32_t PC_GLOBAL *tab = (const int32_t PC_GLOBAL *)(rev_rule ? mp.rc : mp.fw);
        return tab[L]; // -1 == _BAD_OFFSET
    } else if (KIND == 4) { // StratifiedVariableFivePrimeMapFactory.__call__ :765-778
        if (L < mp.min_len || L > mp.max_len || L < 1 || L >= mp.table_len) return -1;
        const int32_t PC_GLOBAL *tab = (const int32_t PC_GLOBAL *)(rev_rule ? mp.rc : mp.fw);
        const int off = tab[L];
        row = L - mp.min_len;
        return off < 0 ? L - 1 : off; // no _BAD_OFFSET check: read_positions[-1]
    }
    return -1;
}

__device__ __forceinline__ int map_kleft_dyn(const MapParams &mp, int L, bool rev_rule, int &row) {
    switch (mp.kind) {
    case 0: return map_kleft<0>(mp, L, rev_rule, row);
    case 1: return map_kleft<1>(mp, L, rev_rule, row);
    case 3: return map_kleft<3>(mp, L, rev_rule, row);
    case 4: return map_kleft<4>(mp, L, rev_rule, row);
    default: row = 0; return -1;
    }
}

// read.positions[k] for a record with aligned runs
__device__ __forceinline__ int32_t walk_runs(const GFile &fv, int64_t i, int nblk, int k) {
    const i32x2 PC_GLOBAL *b = fv.blk + fv.blk_off[i];
    int32_t p = 0;
    for (int j = 0; j < nblk; ++j) {
        i32x2 r = b[j];
        if (k < r.y) { p = r.x + k; break; }
        k -= r.y;
    }
    return p;
}

__device__ __forceinline__ int32_t rec_end(const GFile &fv, int64_t i, int32_t pos, uint32_t meta) {
    // htslib bam_endpos
    int nb = rec_nblk(meta);
    if (nb >= 2) {
        i32x2 r = fv.blk[fv.blk_off[i] + nb - 1];
        return r.x + r.y;
    }
    int L = rec_len(meta);
    return pos + (L > 0 ? L : 1);
}

// ---------------------------------------------------------------- k_tile_ranges
// One thread per (tile, file): the record range a tile has to scan (fetch emulation),
// cut into work items of at most `R` records (load balance for pile-ups), plus the range
// of the file's gapped-record list that can reach the tile.
__device__ __forceinline__ int64_t lower_bound_gap(const u32x4 PC_GLOBAL *rec, int64_t lo, int64_t hi, int64_t key) {
    while (lo < hi) {
        int64_t mid = lo + ((hi - lo) >> 1);
        if ((int64_t)(int32_t)rec[mid].x < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(kWG) void k_tile_ranges(const Tile *__restrict__ tiles, int ntiles,
                                                     const FileView *__restrict__ files, int nfiles,
                                                     int G, int W, int64_t R, WorkItem *work,
                                                     uint32_t *nwork, uint32_t *tile_items,
                                                     uint32_t work_cap) {
    int64_t idx = (int64_t)blockIdx.x * kWG + threadIdx.x;
    if (idx >= (int64_t)ntiles * nfiles) return;
    int t = (int)(idx / nfiles), f = (int)(idx % nfiles);
    Tile tl = tiles[t];
    const GFile fv = gfile(files[f]);
    const int64_t key_lo = (int64_t)tl.win_start - W + 1, key_hi = (int64_t)tl.win_start + G;
    int64_t b0 = fv.tid_bounds[tl.tid], b1 = fv.tid_bounds[tl.tid + 1];
    int64_t lo = lower_bound_pos(fv.rec, b0, b1, key_lo);
    int64_t hi = lower_bound_pos(fv.rec, lo, b1, key_hi);
    int64_t glo = 0, ghi = 0;
    if (fv.ngap) {
        const int64_t g0 = fv.gap_tid_bounds[tl.tid], g1 = fv.gap_tid_bounds[tl.tid + 1];
        glo = lower_bound_gap(fv.gap_rec, g0, g1, key_lo);
        ghi = lower_bound_gap(fv.gap_rec, glo, g1, key_hi);
    }
    int64_t n = hi - lo;
    if (n <= 0 && ghi <= glo) return;
    uint32_t items = n > 0 ? (uint32_t)((n + R - 1) / R) : 1u;
    uint32_t base = atomicAdd(nwork, items);
    atomicAdd(&tile_items[t], items);
    for (uint32_t k = 0; k < items; ++k) {
        if (base + k >= work_cap) break; // cannot happen (capacity is an upper bound); defensive
        WorkItem w;
        w.lo = lo + (int64_t)k * R;
        w.hi = (w.lo + R < hi) ? w.lo + R : hi;
        w.glo = k == 0 ? glo : 0;
        w.ghi = k == 0 ? ghi : 0;
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
// One record -> at most one LDS atomic per strand mode of the tile.  Predicated code, no
// divergent branches; `sbase[m]` is the wave-uniform LDS word offset of mode m's bins, or
// -1 when the tile has no such mode.  `pf`/`pr` = the position the forward / reverse index
// rule selects (valid when kf / kr >= 0).
// AGG: reads arrive sorted by position, so neighbouring lanes often hit the same bin
// (ribosome-profiling pile-ups).  With AGG each maximal run of lanes with the same target
// bin is collapsed into ONE ds_add of the run length (ballot + one ds_bpermute).
template <int KIND, bool AGG>
__device__ __forceinline__ void hist_bin(const MapParams &mp, bool ok, bool rev, int kf, int kr, int32_t pf,
                                         int32_t pr, int row_f, int row_r, const int (&sbase)[kModes],
                                         int32_t win_start, uint32_t G, uint32_t *bins, int lane) {
#pragma unroll
    for (int m = 0; m < kModes; ++m) {
        if (sbase[m] < 0) continue; // wave-uniform
        const bool rr = (m == 1 || m == 3);
        const int k = rr ? kr : kf;
        const uint32_t d = (uint32_t)((rr ? pr : pf) - win_start);
        const bool hit = ok && strand_ok(m, rev) && k >= 0 && d < G;
        const uint32_t addr = (uint32_t)sbase[m] + (KIND == 4 ? (uint32_t)(rr ? row_r : row_f) * G : 0u) + d;
        if (!AGG) {
            if (hit) atomicAdd(&bins[addr], 1u);
        } else {
            const unsigned long long V = __ballot(hit);
            if (V == 0) continue; // wave-uniform
            const unsigned long long lt = (1ull << lane) - 1ull;
            const unsigned long long below = V & lt;
            const int prev = below ? 63 - __clzll(below) : lane;
            const uint32_t pkey = (uint32_t)__shfl((int)addr, prev, 64);
            const bool head = hit && (below == 0 || pkey != addr);
            const unsigned long long H = __ballot(head);
            if (head) {
                const unsigned long long above = H & ~(lt | (1ull << lane));
                const unsigned long long upto = above ? ((above & (0 - above)) - 1ull) : ~0ull; // lanes below the next head
                atomicAdd(&bins[addr], (uint32_t)__popcll(V & upto & ~lt));
            }
        }
    }
}

// ungapped record of the packed stream (gapped ones come from the gapped list, long ones
// from k_long_point)
template <int KIND, bool AGG>
__device__ __forceinline__ void hist_rec(const MapParams &mp, uint32_t rx, uint32_t meta, bool inrange,
                                         const int (&sbase)[kModes], int32_t win_start, uint32_t G,
                                         uint32_t *bins, int lane) {
    const uint32_t fl = rec_flags(meta);
    const int L = rec_len(meta);
    const bool ok = inrange && !(fl & (kFlagExcluded | kFlagLong)) && rec_nblk(meta) < 2 && size_ok(mp, L);
    int row_f, row_r;
    const int kf = map_kleft<KIND>(mp, L, false, row_f);
    const int kr = map_kleft<KIND>(mp, L, true, row_r);
    hist_bin<KIND, AGG>(mp, ok, fl & kFlagReverse, kf, kr, (int32_t)rx + kf, (int32_t)rx + kr, row_f, row_r, sbase,
                        win_start, G, bins, lane);
}

// position of read.positions[k] given the first two runs in registers
__device__ __forceinline__ int32_t walk_from(const GFile &fv, uint32_t off, int nblk, int k, i32x2 b0, i32x2 b1) {
    if (k < b0.y) return b0.x + k;
    k -= b0.y;
    if (k < b1.y) return b1.x + k;
    k -= b1.y;
    int32_t p = 0;
    for (int j = 2; j < nblk; ++j) {
        const i32x2 r = fv.blk[off + j];
        if (k < r.y) { p = r.x + k; break; }
        k -= r.y;
    }
    return p;
}

template <int KIND, bool AGG>
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
    const GFile fv = gfile(files[w.file]);
    const int lane = threadIdx.x & 63;
    int sbase[kModes];
    int nslots = 0;
#pragma unroll
    for (int m = 0; m < kModes; ++m) sbase[m] = ((tl.mode_mask >> m) & 1u) ? (nslots++) * mp.rows * G : -1;
    const int nbins = nslots * mp.rows * G;
    for (int i = threadIdx.x; i < nbins; i += kWG) bins[i] = 0;
    __syncthreads();

    // ---- the packed record stream.  16-byte pairs: one global_load_dwordx4 per lane = 1 KiB
    // per wave instruction, U per lane in flight, and the next batch is requested before the
    // current one is consumed (register double buffer) so HBM latency overlaps the LDS atomics.
    // No dependent loads in this loop.
    constexpr int U = 4;
    const int64_t pair_lo = w.lo >> 1, pair_hi = (w.hi + 1) >> 1;
    const u32x4 none = {0u, kFlagExcluded << 16, 0u, kFlagExcluded << 16};
    u32x4 cur[U], nxt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t j = pair_lo + u * kWG + threadIdx.x;
        cur[u] = (j < pair_hi) ? fv.rec4[j] : none;
    }
    for (int64_t base = pair_lo; base < pair_hi; base += (int64_t)kWG * U) {
        const int64_t nbase = base + (int64_t)kWG * U;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t j = nbase + u * kWG + threadIdx.x;
            nxt[u] = (j < pair_hi) ? fv.rec4[j] : none;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i0 = (base + u * kWG + threadIdx.x) * 2;
            hist_rec<KIND, AGG>(mp, cur[u].x, cur[u].y, i0 >= w.lo && i0 < w.hi, sbase, tl.win_start, (uint32_t)G, bins, lane);
            hist_rec<KIND, AGG>(mp, cur[u].z, cur[u].w, i0 + 1 >= w.lo && i0 + 1 < w.hi, sbase, tl.win_start, (uint32_t)G, bins, lane);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }

    // ---- gapped records (deletions, short introns): their aligned runs live in a side
    // array; consecutive list entries own consecutive runs, so these gathers stay coalesced.
    for (int64_t base = w.glo; base < w.ghi; base += kWG) {
        const int64_t j = base + threadIdx.x;
        const bool in = j < w.ghi;
        const u32x4 g = in ? fv.gap_rec[j] : none;
        const uint32_t meta = g.y, fl = rec_flags(meta);
        const int L = rec_len(meta), nb = rec_nblk(meta);
        const bool ok = in && !(fl & kFlagExcluded) && size_ok(mp, L);
        i32x2 b0 = {0, 1}, b1 = {0, 1};
        if (in) { b0 = fv.blk[g.z]; b1 = fv.blk[g.z + 1]; }
        int row_f, row_r;
        const int kf = map_kleft<KIND>(mp, L, false, row_f);
        const int kr = map_kleft<KIND>(mp, L, true, row_r);
        const int32_t pf = (ok && kf >= 0) ? walk_from(fv, g.z, nb, kf, b0, b1) : 0;
        const int32_t pr = (ok && kr >= 0) ? walk_from(fv, g.z, nb, kr, b0, b1) : 0;
        hist_bin<KIND, AGG>(mp, ok, fl & kFlagReverse, kf, kr, pf, pr, row_f, row_r, sbase, tl.win_start, (uint32_t)G, bins, lane);
    }
    __syncthreads();

    const bool single = tile_items[w.tile] == 1u;
    for (uint32_t pi = tl.piece_begin; pi < tl.piece_end; ++pi) {
        const Piece pc_ = pieces[pi];
        const int rel = pc_.start - tl.win_start;
        for (int r = 0; r < mp.rows; ++r) {
            const uint32_t *src = bins + sbase[pc_.mode] + r * G + rel;
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
                                                    const Piece *__restrict__ pieces, FileView fview,
                                                    MapParams mp, int G, uint32_t plan_modes,
                                                    uint32_t *hist, int64_t hist_row_stride) {
    const GFile fv = gfile(fview);
    int64_t j = (int64_t)blockIdx.x * kWG + threadIdx.x;
    if (j >= fv.nlong) return;
    const int64_t i = fv.long_idx[j];
    const u32x2 r = fv.rec[i];
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
        const int k = map_kleft_dyn(mp, L, m == 1 || m == 3, row);
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
__device__ __forceinline__ void center_one(const GFile &fv, const MapParams &mp, int64_t i, int mode,
                                           const double PC_GLOBAL *inv, int32_t p, double &acc) {
    const u32x2 r = fv.rec[i];
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
        const i32x2 PC_GLOBAL *b = fv.blk + fv.blk_off[i];
        int cum = 0;
        for (int j = 0; j < nb; ++j) {
            const i32x2 run = b[j];
            const int idx = cum + (p - run.x);
            hit |= (p >= run.x) && (p < run.x + run.y) && (idx >= nib) && (idx < L - nib);
            cum += run.y;
        }
    }
    if (hit) acc += val;                    // :254, one IEEE add per covering read, in order
}

__global__ __launch_bounds__(kWG) void k_center(const CenterChunk *__restrict__ chunks, int64_t nchunks,
                                                const FileView *__restrict__ files, int nfiles,
                                                MapParams mp, int W, const double *__restrict__ inv_,
                                                double *hist) {
    const double PC_GLOBAL *inv = (const double PC_GLOBAL *)inv_;
    const int64_t c = __builtin_amdgcn_readfirstlane((int)(((int64_t)blockIdx.x * kWG + threadIdx.x) >> 6));
    if (c >= nchunks) return;
    const int lane = threadIdx.x & 63;
    const CenterChunk ck = chunks[c];
    const int32_t p = ck.start + lane;
    double acc = 0.0;
    for (int f = 0; f < nfiles; ++f) { // file-major, genome_array.py:800-809
        const GFile fv = gfile(files[f]);
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
__global__ __launch_bounds__(kWG) void k_mapped_reads(FileView fview, MapParams mp, int64_t rec_lo, int64_t rec_hi,
                                                      int64_t start, int64_t end, int mode, bool strand_filter,
                                                      uint8_t *mask) {
    const GFile fv = gfile(fview);
    int64_t i = rec_lo + (int64_t)blockIdx.x * kWG + threadIdx.x;
    if (i >= rec_hi) return;
    const u32x2 r = fv.rec[i];
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
            const int k = map_kleft_dyn(mp, L, mode == 1 || mode == 3, row);
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
__global__ __launch_bounds__(kWG) void k_unmappable(FileView fview, MapParams mp, int ntid, Unmappable *list,
                                                    uint32_t cap, uint32_t *count) {
    const GFile fv = gfile(fview);
    int64_t i = (int64_t)blockIdx.x * kWG + threadIdx.x;
    if (i >= fv.n) return;
    const u32x2 r = fv.rec[i];
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
