// plan_kernels.hip.h -- the plan of a large annotation built ON THE GPU (round 4).
//
// pc_plan_create turns the caller's segments (exons of transcripts: plastid/genomics/roitools.pyx SegmentChain,
// counted by get_counts, roitools.pyx:3235-3271) into the tables the counting kernels read: ISLANDS (the union of the
// queried intervals per contig and strand mode), PIECES (islands cut at the window grid), TILES (the pieces of one
// window), OUTPUT PIECES (every queried segment cut at the grid, in the caller's layout, grouped by tile).  The host
// builder (plastid_counts.hip) does that with a dozen threaded passes and sorts: 27 ms for the 479 k exons of a
// human-scale annotation.  Here every pass is a kernel, a radix sort or a scan over the segment table in HBM:
//
//   k_plan_segs      per segment: validation, the per-segment record of the gather pass, its clipped interval as a sort
//                    key (contig | strand mode | start) with the end as the value
//   (radix sort)     intervals by key
//   (scan)           running maximum of the ends within (contig, mode): an interval starts an island iff its start lies
//                    beyond every end before it
//   k_island_*       island table, island lengths -> (scan) -> offsets of the islands in the compact histogram
//   k_seg_island     every segment finds its island (binary search): its place in the compact histogram
//   k_pieces_raw     pieces of every island, keyed (contig | window | mode | offset in the window) -> (radix sort)
//   k_tile_*         tiles = runs of equal (contig, window[, mode]) in the sorted pieces
//   k_out_count/raw  output pieces per segment, each with its tile (binary search) -> (stable radix sort by tile)
//   k_tile_ops       the output-piece range of every tile
//
// The tables are bit-identical to the host builder's (tests/test_gpu_plan.py compares them table by table).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "plastid_counts.h"

namespace pcplan {

using pc::CenterChunk;
using pc::GatherChunk;
using pc::GatherSeg;
using pc::OutPiece;
using pc::Piece;
using pc::Tile;

struct SegIn {
    const int32_t *tid;
    const int64_t *start, *end;
    const uint8_t *strand;
    const int64_t *out_off;
    const int8_t *out_step;
    const int64_t *row_stride;
};

struct Island {
    int32_t tid, mode;
    int64_t s, e, off;
};

// counters and flags the host reads back (one block, three reads)
struct Misc {
    unsigned long long first_bad;   // (segment << 8) | kind of the defect with the lowest segment index; ~0: none
    unsigned long long covered;     // output elements the plan covers
    unsigned long long n_iv;        // clipped, non-empty intervals on known contigs
    unsigned long long npos;        // positions of the compact histogram (sum of the island lengths)
    uint32_t modes, has_sums, needs_zero, max_slots;
    uint32_t n_islands, n_pieces, n_tiles, n_opieces;
    unsigned long long iv_len;      // summed length of those intervals (a sparse annotation takes a smaller window)
};

constexpr int kTidBits = 27;        // contigs a piece key has room for (contig 27 | window 23 | mode 2 | offset 12 bits)

__device__ __forceinline__ int mode_of(uint8_t strand) {   // (plastid_counts.hip mode_of)
    const bool nofilter = strand & PC_STRAND_NOFILTER;
    const int s = strand & 3;
    if (s == PC_STRAND_REV) return nofilter ? 3 : 1;
    if (s == PC_STRAND_FWD) return nofilter ? 2 : 0;
    return 2;
}

// interval key: contig (bits 33..) | mode (31-32) | start (0-30)
__device__ __forceinline__ unsigned long long iv_key(int32_t tid, int mode, int64_t s) {
    return ((unsigned long long)(uint32_t)tid << 33) | ((unsigned long long)mode << 31) | (unsigned long long)s;
}

__global__ __launch_bounds__(256) void k_plan_segs(SegIn in, int64_t nseg, int ntid, int rows, int64_t out_elems, GatherSeg *__restrict__ gsegs,
                                                   unsigned long long *__restrict__ keys, uint32_t *__restrict__ ends, Misc *__restrict__ misc) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long covered = 0, n_iv = 0, iv_len = 0;
    uint32_t modes = 0, sums = 0;
    if (s < nseg) {
        const int64_t st = in.start[s], en = in.end[s], len = en - st;
        const int step = in.out_step[s];
        const int64_t oo = in.out_off[s], rs = in.row_stride[s];
        int bad = 0;
        if (len < 0) bad = 1;
        else if (step != 1 && step != -1 && step != 0) bad = 2;
        else if (len > 0) {
            const int64_t first = oo, last = oo + (int64_t)step * (len - 1);
            const int64_t lo = first < last ? first : last, hi = (first < last ? last : first) + (int64_t)(rows - 1) * rs;
            if (lo < 0 || hi >= out_elems || rs < 0) bad = 3;
        }
        unsigned long long key = ~0ull;
        uint32_t e32 = 0;
        if (bad) atomicMin(&misc->first_bad, ((unsigned long long)s << 8) | (unsigned long long)bad);
        else {
            if (step == 0) sums = 1;
            GatherSeg g;
            g.out_off = oo; g.row_stride = rs; g.len = len; g.step = step; g.pad = 0;
            g.hist_off = -1; g.clip_lo = 0; g.clip_hi = 0; g.start = st;
            covered = (unsigned long long)((step == 0 ? (len > 0 ? 1 : 0) : len) * rows);
            const int32_t t = in.tid[s];
            if (t >= 0 && t < ntid && len != 0) {
                const int64_t cs = st > 0 ? st : 0, ce = en < 0x7fffffffLL ? en : 0x7fffffffLL;
                if (ce > cs) {
                    g.clip_lo = cs - st;
                    g.clip_hi = ce - st;
                    const int m = mode_of(in.strand[s]);
                    modes = 1u << m;
                    key = iv_key(t, m, cs);
                    e32 = (uint32_t)ce;
                    n_iv = 1;
                    iv_len = (unsigned long long)(ce - cs);
                }
            }
            gsegs[s] = g;
        }
        keys[s] = key;
        ends[s] = e32;
    }
    // block totals -> one atomic each
    __shared__ unsigned long long s_cov, s_niv, s_len;
    __shared__ uint32_t s_modes, s_sums;
    if (threadIdx.x == 0) { s_cov = 0; s_niv = 0; s_len = 0; s_modes = 0; s_sums = 0; }
    __syncthreads();
    if (covered) atomicAdd(&s_cov, covered);
    if (n_iv) { atomicAdd(&s_niv, n_iv); atomicAdd(&s_len, iv_len); }
    if (modes) atomicOr(&s_modes, modes);
    if (sums) atomicOr(&s_sums, sums);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_cov) atomicAdd(&misc->covered, s_cov);
        if (s_niv) { atomicAdd(&misc->n_iv, s_niv); atomicAdd(&misc->iv_len, s_len); }
        if (s_modes) atomicOr(&misc->modes, s_modes);
        if (s_sums) atomicOr(&misc->has_sums, s_sums);
    }
}

// running maximum of the ends within (contig, mode): (group << 32 | end), combined in stream order
struct GroupMax {
    __host__ __device__ unsigned long long operator()(unsigned long long a, unsigned long long b) const {
        if ((a >> 32) != (b >> 32)) return b;
        return (uint32_t)a > (uint32_t)b ? a : b;
    }
};

__global__ __launch_bounds__(256) void k_group_ends(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ ends, const Misc *__restrict__ misc,
                                                    unsigned long long *__restrict__ ge) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= misc->n_iv) return;
    ge[i] = ((keys[i] >> 31) << 32) | ends[i];
}

__global__ __launch_bounds__(256) void k_island_flags(const unsigned long long *__restrict__ keys, const unsigned long long *__restrict__ pm, const Misc *__restrict__ misc,
                                                      uint32_t *__restrict__ flags, int64_t n_alloc) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_alloc) return;
    uint32_t f = 0;
    if ((unsigned long long)i < misc->n_iv)
        f = i == 0 || (keys[i] >> 31) != (keys[i - 1] >> 31) || (uint32_t)(keys[i] & 0x7fffffffull) > (uint32_t)pm[i - 1];
    flags[i] = f;
}

// islands from the flagged intervals: start at the flagged element, end = the running maximum at the island's last element
__global__ __launch_bounds__(256) void k_island_fill(const unsigned long long *__restrict__ keys, const unsigned long long *__restrict__ pm, const uint32_t *__restrict__ flags,
                                                     const uint32_t *__restrict__ before, Misc *__restrict__ misc, Island *__restrict__ islands,
                                                     unsigned long long *__restrict__ island_keys) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const unsigned long long n = misc->n_iv;
    if (i >= n) return;
    const uint32_t id = before[i] + flags[i] - 1u;
    if (flags[i]) {
        islands[id].tid = (int32_t)(keys[i] >> 33);
        islands[id].mode = (int32_t)((keys[i] >> 31) & 3u);
        islands[id].s = (int64_t)(keys[i] & 0x7fffffffull);
        island_keys[id] = keys[i];
    }
    if (i + 1 == n || flags[i + 1]) {
        islands[id].e = (int64_t)(uint32_t)pm[i];
        if (i + 1 == n) misc->n_islands = id + 1u;
    }
}

__global__ __launch_bounds__(256) void k_island_lens(const Misc *__restrict__ misc, const Island *__restrict__ islands, int64_t *__restrict__ lens, uint32_t *__restrict__ npieces,
                                                     int G, int64_t n_alloc) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_alloc) return;
    int64_t len = 0;
    uint32_t np = 0;
    if ((uint32_t)i < misc->n_islands) {
        const Island is = islands[i];
        len = is.e - is.s;
        np = is.e > is.s ? (uint32_t)((is.e - 1) / G - is.s / G + 1) : 0u;
    }
    lens[i] = len;
    npieces[i] = np;
}

// island offsets (exclusive sums of the lengths) into the table; totals to the counters
__global__ __launch_bounds__(256) void k_island_offsets(Misc *__restrict__ misc, Island *__restrict__ islands, const int64_t *__restrict__ off, const int64_t *__restrict__ lens,
                                                        const uint32_t *__restrict__ piece_at, const uint32_t *__restrict__ npieces) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t n = misc->n_islands;
    if (i >= n) return;
    islands[i].off = off[i];
    if (i + 1 == n) {
        misc->npos = (unsigned long long)(off[i] + lens[i]);
        misc->n_pieces = piece_at[i] + npieces[i];
    }
}

template <typename K> __device__ __forceinline__ uint32_t upper_bound_u(const K *__restrict__ a, uint32_t n, K v) {   // first index with a[i] > v
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
template <typename K> __device__ __forceinline__ uint32_t lower_bound_u(const K *__restrict__ a, uint32_t n, K v) {   // first index with a[i] >= v
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// every segment -> its island; the output pieces it will make; whether some of its positions lie outside every tile
__global__ __launch_bounds__(256) void k_seg_island(SegIn in, int64_t nseg, GatherSeg *__restrict__ gsegs, Misc *__restrict__ misc, const Island *__restrict__ islands,
                                                    const unsigned long long *__restrict__ island_keys, uint32_t *__restrict__ nout, int G) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= nseg) return;
    GatherSeg g = gsegs[s];
    uint32_t n = 0;
    if (g.clip_hi > g.clip_lo) {
        const int64_t cs = g.start + g.clip_lo, ce = g.start + g.clip_hi;
        const unsigned long long key = iv_key(in.tid[s], mode_of(in.strand[s]), cs);
        const uint32_t lo = upper_bound_u(island_keys, misc->n_islands, key);
        const Island is = islands[lo - 1];
        g.hist_off = is.off + (cs - is.s);
        gsegs[s].hist_off = g.hist_off;
        n = (uint32_t)((ce - 1) / G - cs / G + 1);
    }
    if (g.len > 0 && (g.hist_off < 0 || g.clip_lo > 0 || g.clip_hi < g.len)) atomicOr(&misc->needs_zero, 1u);
    nout[s] = n;
}

// piece key: contig (37..) | window index (14-36) | mode (12-13) | offset in the window (0-11)
__global__ __launch_bounds__(256) void k_pieces_raw(const Misc *__restrict__ misc, const Island *__restrict__ islands, const uint32_t *__restrict__ piece_at, int G,
                                                    unsigned long long *__restrict__ keys, uint32_t *__restrict__ idx, Piece *__restrict__ raw) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= misc->n_pieces) return;
    const uint32_t i = upper_bound_u(piece_at, misc->n_islands, j) - 1u;
    const Island is = islands[i];
    const uint32_t k = j - piece_at[i];
    const int64_t a = k == 0 ? is.s : (is.s / G + (int64_t)k) * G;
    const int64_t win = (a / G) * G;
    const int64_t b = is.e < win + G ? is.e : win + G;
    Piece pc_;
    pc_.hist_off = is.off + (a - is.s); pc_.start = (int32_t)a; pc_.len = (int32_t)(b - a); pc_.mode = is.mode; pc_.pad = 0;
    raw[j] = pc_;
    keys[j] = ((unsigned long long)(uint32_t)is.tid << 37) | ((unsigned long long)(win / G) << 14) | ((unsigned long long)is.mode << 12) | (unsigned long long)(a - win);
    idx[j] = j;
}

__global__ __launch_bounds__(256) void k_pieces_sorted(const Misc *__restrict__ misc, const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ idx,
                                                       const Piece *__restrict__ raw, Piece *__restrict__ pieces, uint32_t *__restrict__ new_tile, int split_modes, int64_t n_alloc) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_alloc) return;
    uint32_t f = 0;
    if ((uint32_t)i < misc->n_pieces) {
        pieces[i] = raw[idx[i]];
        const int sh = split_modes ? 12 : 14;
        f = i == 0 || (keys[i] >> sh) != (keys[i - 1] >> sh);
    }
    new_tile[i] = f;
}

__global__ __launch_bounds__(256) void k_tile_fill(Misc *__restrict__ misc, const unsigned long long *__restrict__ keys, const Piece *__restrict__ pieces,
                                                   const uint32_t *__restrict__ new_tile, const uint32_t *__restrict__ before, int G, int split_modes,
                                                   Tile *__restrict__ tiles, unsigned long long *__restrict__ tile_keys) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t n = misc->n_pieces;
    if (i >= n) return;
    if (i + 1 == n) misc->n_tiles = before[i] + new_tile[i];
    if (!new_tile[i]) return;
    Tile t;
    t.tid = (int32_t)(keys[i] >> 37);
    t.win_start = (int32_t)(((keys[i] >> 14) & 0x7fffffull) * (unsigned long long)G);
    t.piece_begin = i;
    t.mode_mask = 0; t.op_begin = 0; t.op_end = 0;
    uint32_t lo = 0xffffu, hi = 0;
    uint32_t j = i;
    do {
        const Piece pc_ = pieces[j];
        t.mode_mask |= 1u << pc_.mode;
        const uint32_t a = (uint32_t)(pc_.start - t.win_start) & 0xffffu, b = (uint32_t)(pc_.start - t.win_start + pc_.len) & 0xffffu;
        lo = a < lo ? a : lo;
        hi = b > hi ? b : hi;
        ++j;
    } while (j < n && !new_tile[j]);
    t.piece_end = j;
    t.span_lo = (uint16_t)lo; t.span_hi = (uint16_t)hi;
    const uint32_t id = before[i];
    tiles[id] = t;
    // what an output piece looks its tile up by: (contig, window) and -- when every mode has a tile of its own -- the mode
    tile_keys[id] = split_modes ? keys[i] >> 12 : (keys[i] >> 14) << 2;
    if (__popc(t.mode_mask) > 1) atomicMax(&misc->max_slots, (uint32_t)__popc(t.mode_mask));   // (starts at 1: most windows query one strand mode)
}

__global__ __launch_bounds__(256) void k_out_total(Misc *__restrict__ misc, const uint32_t *__restrict__ out_at, const uint32_t *__restrict__ nout, int64_t nseg) {
    if (blockIdx.x == 0 && threadIdx.x == 0) misc->n_opieces = nseg ? out_at[nseg - 1] + nout[nseg - 1] : 0u;
}

__global__ __launch_bounds__(256) void k_out_raw(SegIn in, int64_t nseg, const GatherSeg *__restrict__ gsegs, const Misc *__restrict__ misc,
                                                 const unsigned long long *__restrict__ tile_keys, const uint32_t *__restrict__ out_at, const uint32_t *__restrict__ nout,
                                                 int G, int split_modes, OutPiece *__restrict__ raw, uint32_t *__restrict__ tile_of, uint32_t *__restrict__ idx) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= nseg || nout[s] == 0) return;
    const GatherSeg g = gsegs[s];
    const int m = mode_of(in.strand[s]);
    const int32_t t = in.tid[s];
    const int64_t cs = g.start + g.clip_lo, ce = g.start + g.clip_hi;
    uint32_t at = out_at[s];
    const uint32_t ntiles = misc->n_tiles;
    for (int64_t a = cs; a < ce; ++at) {
        const int64_t win = (a / G) * G;
        const int64_t b = ce < win + G ? ce : win + G;
        const unsigned long long want = ((((unsigned long long)(uint32_t)t << 23) | (unsigned long long)(win / G)) << 2) | (unsigned long long)(split_modes ? m : 0);
        const uint32_t tile = lower_bound_u(tile_keys, ntiles, want);
        OutPiece o;
        o.out_off = g.out_off + (int64_t)g.step * (a - g.start);
        o.row_stride = g.row_stride;
        o.hist_off = g.hist_off + (a - cs);
        o.start = (int32_t)a; o.len = (int32_t)(b - a); o.mode = m; o.step = g.step;
        raw[at] = o;
        tile_of[at] = tile;
        idx[at] = at;
        a = b;
    }
}

__global__ __launch_bounds__(256) void k_out_sorted(const Misc *__restrict__ misc, const uint32_t *__restrict__ tile_of, const uint32_t *__restrict__ idx,
                                                    const OutPiece *__restrict__ raw, OutPiece *__restrict__ opieces, Tile *__restrict__ tiles) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t n = misc->n_opieces;
    if (i >= n) return;
    opieces[i] = raw[idx[i]];
    const uint32_t t = tile_of[i];
    if (i == 0 || tile_of[i - 1] != t) tiles[t].op_begin = i;
    if (i + 1 == n || tile_of[i + 1] != t) tiles[t].op_end = i + 1u;
}

// ---- the tables only the center rule / the coordinate export read, from a GPU-built plan's tables in HBM
// (ensure_center_tables / ensure_gather_tables): 64-position chunks in tile / piece order; (segment, chunk) pairs
__global__ __launch_bounds__(256) void k_cchunk_count(const Tile *__restrict__ tiles, const Piece *__restrict__ pieces, uint32_t ntiles, uint32_t *__restrict__ n) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= ntiles) return;
    uint32_t k = 0;
    for (uint32_t i = tiles[t].piece_begin; i < tiles[t].piece_end; ++i) k += (uint32_t)(pieces[i].len + pc::kWave - 1) / pc::kWave;
    n[t] = k;
}

__global__ __launch_bounds__(256) void k_cchunk_fill(const Tile *__restrict__ tiles, const Piece *__restrict__ pieces, uint32_t ntiles, const uint32_t *__restrict__ at,
                                                     CenterChunk *__restrict__ chunks) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= ntiles) return;
    const Tile tl = tiles[t];
    uint32_t k = at[t];
    for (uint32_t i = tl.piece_begin; i < tl.piece_end; ++i) {
        const Piece pc_ = pieces[i];
        for (int32_t a = 0; a < pc_.len; a += pc::kWave) {
            CenterChunk c;
            c.hist_off = pc_.hist_off + a; c.tid = tl.tid; c.start = pc_.start + a;
            c.len = pc_.len - a < pc::kWave ? pc_.len - a : pc::kWave; c.mode = pc_.mode;
            c.op_begin = tl.op_begin; c.op_end = tl.op_end;
            chunks[k++] = c;
        }
    }
}

__global__ __launch_bounds__(256) void k_gchunk_count(const GatherSeg *__restrict__ gsegs, int64_t nseg, uint32_t *__restrict__ n) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= nseg) return;
    n[s] = (uint32_t)((gsegs[s].len + pc::kGatherChunk - 1) / pc::kGatherChunk);
}

__global__ __launch_bounds__(256) void k_gchunk_fill(int64_t nseg, const uint32_t *__restrict__ n, const uint32_t *__restrict__ at, GatherChunk *__restrict__ chunks) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= nseg) return;
    const uint32_t base = at[s];
    for (uint32_t c = 0; c < n[s]; ++c) { chunks[base + c].seg = (uint32_t)s; chunks[base + c].chunk = c; }
}

} // namespace pcplan
