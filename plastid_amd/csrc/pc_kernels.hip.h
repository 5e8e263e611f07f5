// pc_kernels.hip.h -- device code of the MI355X per-position read-counting engine.
//
// gfx950 only (wave64, 256 CUs in 8 XCDs, 160 KiB LDS/CU, HBM3E).  This path is
// integer/byte scatter-reduce work bound by HBM bandwidth: no MFMA anywhere.
//
// Data layout in HBM (per staged file, BAM order; DESIGN.md section 3 has the full table)
//   stream[i]   uint32 4-byte record the tile kernel streams: low half of pos | L | strand | skip
//   rec[i]      uint2 {pos:int32, meta:uint32}, meta = L | flags<<16 | nblk<<24   (center /
//               reads_out / warning kernels)
//   gap_rec, long_rec (+ *_runs)   side lists: gapped / over-long and long-span records with their
//               first two aligned runs; blk_off[i], blk[j] int2 {start,len}: all runs of nblk>=2 records
//   lin_tab, glin_tab, llin_tab, plin_tab   linear index (first record at/after every 128-nt bucket)
//   hist        compact coverage over the *union* of queried intervals per strand mode
//               ("islands"): uint32, what merged windows of the point rules go through
//   out         the caller-visible int64/float64 vectors (every chain 5'->3')
//
// Kernels (reference = plastid/genomics/map_factories.pyx unless noted)
//   k_tile_ranges   fetch emulation: record range of every genome window  (genome_array.py:800-809)
//   k_hist_point    FivePrime/ThreePrime/Variable/Stratified + filters + get_counts layout
//                                                                         (:308-367,:407-466,:585-650,:724-780)
//   k_gather_split  lays out windows that were split into several work items
//   k_cs_count / k_cs_scatter   center streams: the aligned runs of the reads a strand selection keeps, record order
//   k_center_weigh / k_center_order / k_center   CenterMapFactory, ordered float64 replay (:200-265)
//                   + SegmentChain.get_counts layout + normalisation    (roitools.pyx:3259-3271, genome_array.py:826-830)
//   k_rle_*         run-length encoding of an output vector (export, genome_array.py:990-1111)
//   k_total_*       sum of an output vector (multi-GPU summary totals)
//   k_mapped_reads  reads_out of the map functions for one segment
//   k_unmappable    records that make the reference emit its DataWarning
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The kernels use gfx950 encodings directly (64-bit DPP with row_newbcast in k_center's replay, fixed wave64 layouts):
// any other offload architecture is rejected here rather than at assembly time.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "plastid_amd is written for gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif

namespace pc {

constexpr int kWG = 256;          // 4 waves of 64
#ifndef PC_HIST_WG
#define PC_HIST_WG 256            // workgroup size of the histogram kernel
#endif
// Waves per SIMD the tile kernel is compiled for (register budget 512 / waves), and 16-byte loads in flight per lane
// (x2: register double buffer).  Sparse windows are latency-bound, so residency counts there: every instantiation
// fits six waves (76 - 77 VGPRs) and, without the first-batch prefetch of the gapped-record list, seven (66 - 69)
// without scratch; eight spill.  Measured (scripts/exp_hist_occupancy.sh): C4 (variable offsets) 1.51 ms at five
// waves, 1.36 at six, 1.22 at seven; C2 (dense, HBM-bound) 0.155 at six, 0.158 at seven; C5 (stratified: its
// 11-row window allowed six workgroups per CU by LDS alone then) 5.0 at six, 5.1 at seven.  Round 3: the run stream goes
// through the entry table (every instantiation fits 66 - 69 VGPRs at seven waves, no scratch) and a multi-row plan gives
// every strand mode of a window its own tile, i.e. half the LDS -- so seven for the variable-offset and the stratified
// rule, six for the dense single-row rules.
#ifndef PC_HIST_WAVES
#define PC_HIST_WAVES(KIND) ((KIND) >= 3 ? 7 : 6)
#endif
#ifndef PC_HIST_U
#define PC_HIST_U(KIND) 4
#endif
constexpr int kHistWG = PC_HIST_WG;
constexpr int kWave = 64;
constexpr uint32_t kFlagReverse = 0x01;
constexpr uint32_t kFlagUser = 0x08;      // engine-internal: excluded by the CALLER's own filters (the PC_FLAG_EXCLUDED it staged); kFlagExcluded = this OR the verdict of the FLAG / MAPQ filter (pc_set_flag_filter)
constexpr uint32_t kFlagWide = 0x10;      // engine-internal: aligned length > 65 535 or > 255 runs -- the 16 / 8-bit fields read 65535 / 255, true values aside
constexpr uint32_t kFlagRuns = 0x20;      // engine-internal: every aligned run of the record is in the run stream
constexpr uint32_t kFlagLong = 0x40;      // engine-internal: span > W, handled by the long-read path
constexpr uint32_t kFlagExcluded = 0x80;
// the flag byte of a staged record from the caller's PC_FLAG_* bits
__host__ __device__ inline uint32_t caller_flags(uint32_t f) { return (f & kFlagReverse) | ((f & kFlagExcluded) ? (kFlagExcluded | kFlagUser) : 0u); }
constexpr int kGatherChunk = 1024;
constexpr int kLinShift = 7;              // linear-index bucket = 128 genome positions
constexpr int kStreamMaxLen = 255;        // aligned lengths the 4-byte record stream can carry

// The record stream the tile kernel reads is 4 bytes per record:
//   bit  0      skip: excluded by a host-side filter, or binned from a side list instead (gapped,
//               long-span, or longer than kStreamMaxLen)
//   bit  2      reverse strand
//   bits 4-11   aligned length L (<= kStreamMaxLen)
//   bits 16-31  low half of pos (the high half is implied by the window: every record a window
//               scans lies within +-32 K positions of its start)
// The fields sit where the kernel wants them: `word & 0xff4` is the byte offset of the record's
// (length, strand) entry in the LDS table, `(word + entry) >> 16` its window-relative position.
constexpr uint32_t kStreamSkip = 1u;
__host__ __device__ inline uint32_t stream_word(uint32_t pos, uint32_t meta) {
    const uint32_t L = meta & 0xffffu, fl = (meta >> 16) & 0xffu, nb = meta >> 24;
    const bool skip = (fl & (kFlagExcluded | kFlagLong)) != 0u || nb >= 2u || L > (uint32_t)kStreamMaxLen;
    return (pos << 16) | (skip ? kStreamSkip : (L << 4)) | ((fl & kFlagReverse) << 2);
}
__host__ __device__ inline uint32_t stream_len(uint32_t word) { return (word >> 4) & 0xffu; }

// strand modes of a query interval
//   0: '+'  keeps forward reads, forward index rule
//   1: '-'  keeps reverse reads, reverse index rule
//   2: '.'  keeps all reads,     forward index rule  (map_factories.pyx:345-346: only '-' flips)
//   3: all reads, reverse index rule (direct map-factory call on a '-' segment, no strand filter)
constexpr int kModes = 4;

struct FileView {
    const uint2 *rec;
    const uint32_t *stream;         // 4-byte record stream (see stream_word), padded to a multiple of 4 with skip words
    const uint32_t *blk_off;
    const int2 *blk;
    const int64_t *tid_bounds;      // ntid+1
    const uint32_t *long_idx;       // long-span records, record order
    const int32_t *long_tid;
    const int32_t *long_pmax;       // prefix max of ref_end within a tid
    const int64_t *long_tid_bounds; // ntid+1
    const uint4 *long_rec;          // long-span records {pos, meta, blk_off, rec_idx}, record order
    const uint4 *gap_rec;           // short-span gapped records {pos, meta, blk_off, rec_idx}, record order
    const int4 *gap_runs;           // their first two aligned runs {start0, len0, start1, len1} (no dependent load)
    const int4 *long_runs;          // same for the long-span list
    const int64_t *gap_tid_bounds;  // ntid+1
    // linear index (cf. the BAI linear index): first record at or after every kLinShift-bit
    // genome bucket, per contig; lin_off[t] = start of contig t's buckets (nb_t + 1 entries)
    const uint32_t *lin_tab;
    const uint32_t *glin_tab;       // same for the gapped-record list
    const uint32_t *llin_tab;       // same for the long-span list (first long read at/after the bucket)
    const uint32_t *plin_tab;       // first long read whose running max end exceeds the bucket start
    const int64_t *lin_off;         // ntid+1 (shared by both tables)
    // run stream: one 8-byte record per aligned run of every multi-run read with L <= kStreamMaxLen, sorted
    // by (contig, run start) -- what the point rules scan instead of the gapped / long-span lists
    const uint2 *run_rec;           // {run start, run length | read index of the run's first base << 8 | L << 16 | flags << 24}
    const uint32_t *rlin_tab;       // first run at/after every bucket
    // long-span reads that are NOT in the run stream (aligned length > kStreamMaxLen): the point rules' own list
    const uint4 *xlong_rec;
    const int4 *xlong_runs;
    const uint32_t *xllin_tab;
    const uint32_t *xplin_tab;
    // center streams (see k_center): entries and entries-before-record per strand selection (forward / reverse / all reads)
    const uint2 *cs_ent[3];
    const uint32_t *cs_soff[3];
    uint32_t cs_total[3];           // entries of each stream (64 entries that cover nothing follow them)
    uint32_t cs_indirect[3];        // 1: the stream holds indirect entries (reads beyond the 8-bit fields)
    // wide records: true {aligned length, run count} per long-list entry (nullptr: the file has none) and by record index
    const uint2 *long_wide;
    const uint2 *xlong_wide;
    const uint32_t *wide_rec;
    const uint2 *wide_val;
    int64_t nwide;
    int64_t n;
    int64_t nlong;
    int64_t ngap;
    int64_t nrunrec;
    int64_t nxlong;
};

// Pointers that come out of a FileView are loaded from memory, so the compiler only knows
// them as generic ("flat") pointers: flat loads count on both vmcnt and lgkmcnt and cannot be
// scalarised.  GFile re-types them as global (address space 1) once per kernel.
#define PC_GLOBAL __attribute__((address_space(1)))
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct GFile {
    const u32x2 PC_GLOBAL *rec;
    const u32x4 PC_GLOBAL *stream4; // the 4-byte record stream, four records per 16-byte load
    const uint32_t PC_GLOBAL *blk_off;
    const i32x2 PC_GLOBAL *blk;
    const int64_t PC_GLOBAL *tid_bounds;
    const uint32_t PC_GLOBAL *long_idx;
    const int32_t PC_GLOBAL *long_tid;
    const int32_t PC_GLOBAL *long_pmax;
    const int64_t PC_GLOBAL *long_tid_bounds;
    const u32x4 PC_GLOBAL *long_rec;
    const u32x4 PC_GLOBAL *gap_rec;
    const i32x4 PC_GLOBAL *gap_runs;
    const i32x4 PC_GLOBAL *long_runs;
    const int64_t PC_GLOBAL *gap_tid_bounds;
    const uint32_t PC_GLOBAL *lin_tab;
    const uint32_t PC_GLOBAL *glin_tab;
    const uint32_t PC_GLOBAL *llin_tab;
    const uint32_t PC_GLOBAL *plin_tab;
    const int64_t PC_GLOBAL *lin_off;
    const u32x2 PC_GLOBAL *run_rec;
    const uint32_t PC_GLOBAL *rlin_tab;
    const u32x4 PC_GLOBAL *xlong_rec;
    const i32x4 PC_GLOBAL *xlong_runs;
    const uint32_t PC_GLOBAL *xllin_tab;
    const uint32_t PC_GLOBAL *xplin_tab;
    const u32x2 PC_GLOBAL *long_wide;
    const u32x2 PC_GLOBAL *xlong_wide;
    const uint32_t PC_GLOBAL *wide_rec;
    const u32x2 PC_GLOBAL *wide_val;
    int64_t nwide;
    int64_t n;
    int64_t nlong;
    int64_t ngap;
    int64_t nrunrec;
    int64_t nxlong;
};

__device__ __forceinline__ GFile gfile(const FileView &v) {
    GFile g;
    g.rec = (const u32x2 PC_GLOBAL *)v.rec;
    g.stream4 = (const u32x4 PC_GLOBAL *)v.stream;
    g.blk_off = (const uint32_t PC_GLOBAL *)v.blk_off;
    g.blk = (const i32x2 PC_GLOBAL *)v.blk;
    g.tid_bounds = (const int64_t PC_GLOBAL *)v.tid_bounds;
    g.long_idx = (const uint32_t PC_GLOBAL *)v.long_idx;
    g.long_tid = (const int32_t PC_GLOBAL *)v.long_tid;
    g.long_pmax = (const int32_t PC_GLOBAL *)v.long_pmax;
    g.long_tid_bounds = (const int64_t PC_GLOBAL *)v.long_tid_bounds;
    g.long_rec = (const u32x4 PC_GLOBAL *)v.long_rec;
    g.gap_rec = (const u32x4 PC_GLOBAL *)v.gap_rec;
    g.gap_runs = (const i32x4 PC_GLOBAL *)v.gap_runs;
    g.long_runs = (const i32x4 PC_GLOBAL *)v.long_runs;
    g.gap_tid_bounds = (const int64_t PC_GLOBAL *)v.gap_tid_bounds;
    g.lin_tab = (const uint32_t PC_GLOBAL *)v.lin_tab;
    g.glin_tab = (const uint32_t PC_GLOBAL *)v.glin_tab;
    g.llin_tab = (const uint32_t PC_GLOBAL *)v.llin_tab;
    g.plin_tab = (const uint32_t PC_GLOBAL *)v.plin_tab;
    g.lin_off = (const int64_t PC_GLOBAL *)v.lin_off;
    g.run_rec = (const u32x2 PC_GLOBAL *)v.run_rec;
    g.rlin_tab = (const uint32_t PC_GLOBAL *)v.rlin_tab;
    g.xlong_rec = (const u32x4 PC_GLOBAL *)v.xlong_rec;
    g.xlong_runs = (const i32x4 PC_GLOBAL *)v.xlong_runs;
    g.xllin_tab = (const uint32_t PC_GLOBAL *)v.xllin_tab;
    g.xplin_tab = (const uint32_t PC_GLOBAL *)v.xplin_tab;
    // (the center streams are picked by a run-time index: read from the FileView in memory, cs_stream)
    g.long_wide = (const u32x2 PC_GLOBAL *)v.long_wide;
    g.xlong_wide = (const u32x2 PC_GLOBAL *)v.xlong_wide;
    g.wide_rec = (const uint32_t PC_GLOBAL *)v.wide_rec;
    g.wide_val = (const u32x2 PC_GLOBAL *)v.wide_val;
    g.nwide = v.nwide;
    g.n = v.n;
    g.nlong = v.nlong;
    g.ngap = v.ngap;
    g.nrunrec = v.nrunrec;
    g.nxlong = v.nxlong;
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
    uint32_t piece_begin; // island pieces (histogram coordinates)
    uint32_t piece_end;
    uint32_t mode_mask;
    uint32_t op_begin;    // output pieces (segment slices in the caller's layout)
    uint32_t op_end;
    uint16_t span_lo;     // queried positions of the window all lie in [span_lo, span_hi) (window-relative)
    uint16_t span_hi;
};

// A queried segment cut at the tile grid, with its place in the caller's output buffer:
// position start+i, row r  ->  out[out_off + step*i + r*row_stride]
struct OutPiece {
    int64_t out_off;
    int64_t row_stride;
    int64_t hist_off; // same positions in the compact histogram (used when a tile is split)
    int32_t start;
    int32_t len;
    int32_t mode;
    int32_t step;
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
    int64_t llo, lhi; // candidate range of the long-span list (first work item only)
    uint32_t rlo, rhi; // range of the run stream
    uint32_t tile;
    uint32_t file;
    int32_t win_start; // copied from the tile: saves the histogram kernel a dependent load
    uint32_t mode_mask;
    uint32_t piece_begin, piece_end;
    uint32_t op_begin, op_end;
    int32_t sub_lo, sub_hi; // the part of the window this item owns (window-relative positions)
    uint32_t merge;         // 1: several items share the window (pile-up / several files) -> merge via hist
    uint16_t span_lo, span_hi; // bins that can ever be read back (see Tile)
};



// Several alignment files: one work item serves a window for ALL of them (joint window).  The item itself carries the
// ranges of file 0; those of file f >= 1 sit in a side array at [slot * (nfiles - 1) + f - 1].
struct FileRange {
    int64_t lo, hi, glo, ghi, llo, lhi;
    uint32_t rlo, rhi;
};

struct CenterChunk {
    int64_t hist_off;
    int32_t tid;
    int32_t start;
    int32_t len;
    int32_t mode;
    uint32_t op_begin, op_end;   // output pieces of the chunk's window (Tile::op_begin / op_end): where its sums go
};

struct GatherSeg {
    int64_t out_off;
    int64_t row_stride;
    int64_t hist_off; // hist index of position (start + clip_lo); -1: all zero
    int64_t len;
    int64_t clip_lo, clip_hi;
    int64_t start;    // genomic coordinate of the segment's first position
    int32_t step;
    int32_t pad;
};

struct GatherChunk {
    uint32_t seg;
    uint32_t chunk;
};

struct Unmappable {
    int32_t tid, pos, end, rev;
    int32_t len;      // aligned length (the reference names it in the Variable rule's warning, :633-648)
    uint32_t rec;     // record index in its file (fetch order: file-major, then record order)
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

// same, with the offset-table entry of (L, rule) already in a register (`tabval`)
template <int KIND>
__device__ __forceinline__ int map_kleft_val(const MapParams &mp, int L, bool rev_rule, int &row, int tabval) {
    if (KIND == 3) {
        row = 0;
        return L >= mp.table_len ? -1 : tabval;
    } else if (KIND == 4) {
        row = 0;
        if (L < mp.min_len || L > mp.max_len || L < 1 || L >= mp.table_len) return -1;
        row = L - mp.min_len;
        return tabval < 0 ? L - 1 : tabval;
    }
    return map_kleft<KIND>(mp, L, rev_rule, row);
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

// true {aligned length, run count} of record i: the packed header, or -- for a wide record -- the side table
__device__ __forceinline__ void rec_true(const GFile &fv, int64_t i, uint32_t meta, int &L, int &nb) {
    L = rec_len(meta);
    nb = rec_nblk(meta);
    if (rec_flags(meta) & kFlagWide) {
        int64_t lo = 0, hi = fv.nwide;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)fv.wide_rec[mid] < i) lo = mid + 1; else hi = mid;
        }
        const u32x2 v = fv.wide_val[lo];
        L = (int)v.x;
        nb = (int)v.y;
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
    int L, nb;
    rec_true(fv, i, meta, L, nb);
    if (nb >= 2) {
        i32x2 r = fv.blk[fv.blk_off[i] + nb - 1];
        return r.x + r.y;
    }
    return pos + (L > 0 ? L : 1);
}

// ---------------------------------------------------------------- k_tile_ranges
// One WAVE per (tile, file): the record range a tile has to scan (fetch emulation), cut
// into work items of at most `R` records (load balance for pile-ups), plus the range of
// the file's gapped-record list that can reach the tile.  Each search starts from the
// linear index (one lookup narrows it to a 128-nt bucket) and finishes 64-ary: every step
// probes 64 evenly spaced records of the bucket at once (ballot), so a search costs about
// three dependent loads instead of the ~27 of a binary search over the whole contig.
template <int STRIDE> // dwords between consecutive records: 2 (packed stream) or 4 (gapped list)
__device__ __forceinline__ int64_t wave_lower_bound(const uint32_t PC_GLOBAL *pos, int64_t lo, int64_t hi, int64_t key,
                                                    int lane) {
    while (hi - lo > 64) {
        const int64_t step = (hi - lo + 63) >> 6;
        int64_t idx = lo + (int64_t)(lane + 1) * step - 1;
        if (idx >= hi) idx = hi - 1;
        const bool less = (int64_t)(int32_t)pos[idx * STRIDE] < key;
        const int c = __popcll(__ballot(less)); // probes are monotone: the first c are < key
        const int64_t nlo = c == 0 ? lo : lo + (int64_t)c * step;          // one past probe c-1
        const int64_t nhi = c == 64 ? hi : lo + (int64_t)(c + 1) * step - 1; // probe c
        lo = nlo < hi ? nlo : hi;
        hi = nhi < hi ? nhi : hi;
        if (hi < lo) hi = lo;
    }
    const int64_t idx = lo + lane;
    const bool less = idx < hi && (int64_t)(int32_t)pos[idx * STRIDE] < key;
    return lo + __popcll(__ballot(less));
}

template <int STRIDE>
__device__ __forceinline__ int64_t indexed_lower_bound(const uint32_t PC_GLOBAL *pos, const uint32_t PC_GLOBAL *lin,
                                                       int64_t lin0, int64_t nb, int64_t key, int lane) {
    // lin[lin0 + b] = first record with pos >= b << kLinShift, b = 0..nb (entry nb = contig end)
    int64_t b = key <= 0 ? 0 : (key >> kLinShift);
    if (b > nb) b = nb;
    const int64_t b1 = b + 1 > nb ? nb : b + 1;
    const int64_t lo = lin[lin0 + b], hi = lin[lin0 + b1];
    return wave_lower_bound<STRIDE>(pos, lo, hi, key, lane);
}

constexpr int kRangesWG = 256;
constexpr int kMaxSub = 32;     // a dense window is cut into up to 32 sub-windows (>= 128 positions each)

// linear-index lookup: first record at/after the bucket holding `key` (conservative: rounds down)
__device__ __forceinline__ int64_t lin_floor(const uint32_t PC_GLOBAL *lin, int64_t lin0, int64_t nb, int64_t key) {
    int64_t b = key <= 0 ? 0 : (key >> kLinShift);
    if (b > nb) b = nb;
    return lin[lin0 + b];
}

// The same lookup made EXACT: first entry of the contig whose start is >= key -- the bucket of `key`, then a bisection
// inside it.  `v`: the sorted starts as a stride-STRIDE dword array (packed records: 2, run stream: 2, side lists: 4).
// Short windows use it (k_tile_ranges): a 256-position window whose lower record bound is rounded down to a 128-nt
// bucket scans up to half again as many records as it bins (C5: 6.96 GB read for a 4.3 GB stream in round 4).
template <int STRIDE>
__device__ __forceinline__ int64_t lin_exact(const uint32_t PC_GLOBAL *v, const uint32_t PC_GLOBAL *lin, int64_t lin0, int64_t nb, int64_t key) {
    int64_t b = key <= 0 ? 0 : (key >> kLinShift);
    if (b > nb) b = nb;
    const int64_t b1 = b + 1 > nb ? nb : b + 1;
    int64_t lo = lin[lin0 + b], hi = lin[lin0 + b1];
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if ((int64_t)(int32_t)v[mid * STRIDE] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// Up to four exact lookups at once (k_tile_ranges: both bounds of the record stream and of the run stream of one
// window).  The bucket edges of all of them are requested together, and every round of the searches issues one load per
// search before it looks at any of them -- a thread that ran them one after the other waited for ~40 dependent loads.
// CG = 1: plain bisections, interleaved.  CG = 16: the sixteen lanes `sub` = 0..15 of an aligned group work on the SAME
// window and every round probes sixteen evenly spaced entries (three rounds for a bucket of 4 096 entries instead of
// twelve): what small plans use, whose few thousand windows would otherwise leave most of the chip idle.
// key[k] / v[k] / lin[k]: the searches; n: how many of them are live.  Returns lower bounds as lin_exact does.
template <int CG>
__device__ __forceinline__ void lin_exact_multi(const uint32_t PC_GLOBAL *const (&v)[4], const uint32_t PC_GLOBAL *const (&lin)[4],
                                                int64_t lin0, int64_t nb, const int64_t (&key)[4], int n, int sub, int64_t (&out)[4]) {
    int64_t lo[4], hi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        lo[k] = hi[k] = 0;
        if (k < n) {
            int64_t b = key[k] <= 0 ? 0 : (key[k] >> kLinShift);
            if (b > nb) b = nb;
            const int64_t b1 = b + 1 > nb ? nb : b + 1;
            lo[k] = lin[k][lin0 + b];
            hi[k] = lin[k][lin0 + b1];
        }
    }
    if (CG == 1) {
        while ((lo[0] < hi[0]) | (lo[1] < hi[1]) | (lo[2] < hi[2]) | (lo[3] < hi[3])) {
            int64_t mid[4];
            int32_t val[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                mid[k] = lo[k] + ((hi[k] - lo[k]) >> 1);
                val[k] = lo[k] < hi[k] ? (int32_t)v[k][mid[k] * 2] : 0;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (lo[k] < hi[k]) { if ((int64_t)val[k] < key[k]) lo[k] = mid[k] + 1; else hi[k] = mid[k]; }
        }
    } else {
        const int gbase = (int)(threadIdx.x & 63u) & ~(CG - 1);   // first lane of this group inside its wave
        const unsigned long long gmask = CG >= 64 ? ~0ull : ((1ull << CG) - 1ull);
        // (every lane of a group holds the same lo / hi: the loop conditions are uniform inside the group, and a ballot
        // only ever looks at the group's own bits)
        while ((hi[0] - lo[0] > CG) | (hi[1] - lo[1] > CG) | (hi[2] - lo[2] > CG) | (hi[3] - lo[3] > CG)) {
            int64_t step[4];
            int32_t val[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                step[k] = (hi[k] - lo[k] + CG - 1) / CG;
                int64_t idx = lo[k] + (int64_t)(sub + 1) * step[k] - 1;
                if (idx >= hi[k]) idx = hi[k] - 1;
                val[k] = hi[k] - lo[k] > CG ? (int32_t)v[k][idx * 2] : 0;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool wide = hi[k] - lo[k] > CG;
                const unsigned long long bal = (__ballot(wide && (int64_t)val[k] < key[k]) >> gbase) & gmask;
                if (wide) {   // probes are monotone: the first c are < key
                    const int c = __popcll(bal);
                    const int64_t nlo = c == 0 ? lo[k] : lo[k] + (int64_t)c * step[k];            // one past probe c - 1
                    const int64_t nhi = c == CG ? hi[k] : lo[k] + (int64_t)(c + 1) * step[k] - 1;  // probe c
                    const int64_t h0 = hi[k];
                    lo[k] = nlo < h0 ? nlo : h0;
                    hi[k] = nhi < h0 ? nhi : h0;
                    if (hi[k] < lo[k]) hi[k] = lo[k];
                }
            }
        }
        int32_t val[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) val[k] = lo[k] + sub < hi[k] ? (int32_t)v[k][(lo[k] + sub) * 2] : 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned long long bal = (__ballot(lo[k] + sub < hi[k] && (int64_t)val[k] < key[k]) >> gbase) & gmask;
            lo[k] += __popcll(bal);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = lo[k];
}

// Test hook (never set in the product build): a variant compiled with -DPC_KMAX16=1000000000 has NO guard on the 16-bit
// bins of the stratified rule -- tests/test_gpu_parity.py::test_sixteen_bit_bins_do_not_overflow must fail on it.
#ifndef PC_KMAX16
#define PC_KMAX16 65535
#endif
// (spans of at most this many positions get exact bounds: beyond it the bucket rounding is a few percent of the scan)
constexpr int kExactSpan = 1024;

// exclusive prefix sum of one value per thread over a kRangesWG-thread block
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, uint32_t *s_wave, uint32_t &total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) s_wave[wv] = x;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < kRangesWG / 64; ++i) {
        const uint32_t t = s_wave[i];
        if (i < wv) base += t;
        tot += t;
    }
    total = tot;
    __syncthreads();
    return base + x - v;
}

__device__ __forceinline__ unsigned long long block_scan_excl64(unsigned long long v, unsigned long long *s_wave,
                                                               unsigned long long &total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) s_wave[wv] = x;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < kRangesWG / 64; ++i) {
        const unsigned long long t = s_wave[i];
        if (i < wv) base += t;
        tot += t;
    }
    total = tot;
    return base + x - v;
}

// first index in [lo,hi) of a stride-`STRIDE` dword array whose value >= key (per-thread bisection)
template <int STRIDE>
__device__ __forceinline__ int64_t lower_bound_i32(const uint32_t PC_GLOBAL *v, int64_t lo, int64_t hi, int64_t key) {
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if ((int64_t)(int32_t)v[mid * STRIDE] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// One THREAD per (tile, file): which records the tile must scan (fetch emulation,
// genome_array.py:800-809).  Two halos: the 4-byte stream only carries single-run reads, which reach
// at most `Ws` (their longest aligned length) positions to the right of their start, so the stream
// scan starts Ws before the first queried position; gapped reads with a short span come from the side
// list, whose scan starts `W` (their longest span) before it.  On spliced data W is ~30x Ws, and a
// stream scan with the wide halo would read every record several times over.  Window edges are multiples of the 128-nt linear-index bucket, so
// the upper ends are exact table lookups and the lower ends are rounded down to a bucket (a
// few extra records are streamed; they fall outside the bins) -- no searching at all for the
// packed stream.  A dense window is cut into sub-windows, each an independent work item that
// owns its slice of the output -- no merging.  Only when a sub-window alone holds a pile-up
// (> `pile` records), or several files feed one window, the window falls back to record slices
// merged through the compact histogram.
//
// Work-list slots, three classes:
//   heavy  items that scan more than R records (dense sub-windows): queued from the FRONT of the
//          list -- the histogram kernel dispatches front first, so the long items start at t = 0
//          and the short ones fill the tail (longest-processing-time-first, in two classes);
//   light  everything else, queued from the BACK;
//   small  sparse windows (all queried positions within `small_g`, few records): their own list,
//          served by single-wave workgroups with a small LDS footprint -- a sparse annotation is
//          latency-bound, so what matters is how many windows are in flight per CU.
// One returning atomic per class per workgroup: a single hot counter saturates near 90/us.
template <int CG>
__global__ __launch_bounds__(kRangesWG) void k_tile_ranges(const Tile *__restrict__ tiles, int ntiles,
                                                           FileView file0, const FileView *__restrict__ files,
                                                           int nfiles, int G, int W, int Ws, int Wr, int64_t R, int64_t pile,
                                                           WorkItem *work, uint32_t *nwork, uint32_t *tile_items,
                                                           uint32_t work_cap, WorkItem *work_small, int small_g,
                                                           int64_t small_n, int diag, FileRange *chain, FileRange *chain_small, int b16) {
    // b16: the plan's windows are binned into 16-BIT counters (several rows: the stratified rule, k_hist_point) -- no work
    // item may add 65 536 times or more to one bin.  An item's adds are bounded by what it scans (records, runs, list
    // entries), so: a window is cut into sub-windows by that total, one whose (sub-)window still scans more than 65 535
    // is merged through the compact histogram, and a merged window's slices hold ONE kind of range each, at most R
    // (49 152) entries of it.
    constexpr int64_t kMax16 = PC_KMAX16;
    __shared__ unsigned long long s_wave64[kRangesWG / 64];
    __shared__ uint32_t s_base[3];
    // CG lanes per window (1, or 16 for small plans: the exact bounds are then searched by the group, lin_exact_multi);
    // the lanes of a group do everything else alike, and lane 0 of the group queues and writes
    const int64_t gtid = (int64_t)blockIdx.x * kRangesWG + threadIdx.x;
    const int64_t idx = gtid / CG;
    const int gsub = (int)(gtid % CG);
    // several files: ONE thread per window looks at all of them, and the window's work items are joint (every file's
    // records binned into the same LDS bins, the output written once) -- only a pile-up still merges through the
    // compact histogram
    const bool joint = nfiles > 1;
    const bool live = idx < (joint ? (int64_t)ntiles : (int64_t)ntiles * nfiles);
    int t = 0, f = 0, S = 1;
    Tile tl = {};
    GFile fv = {};
    int64_t l0 = 0, nb = 0, ws = 0;
    int64_t wlo = 0, whi = 0, wglo = 0, wghi = 0, llo = 0, lhi = 0, wrlo = 0, wrhi = 0;
    uint32_t n_heavy = 0, n_light = 0, n_small = 0;
    bool merge = false;
    // CG == 16, one file, a window cut into sub-windows: the sixteen lanes of the group share the sub-windows (lane g looks
    // at sub-windows g and g + 16: their bounds are a dozen dependent index reads each, which one lane would do one
    // sub-window after the other -- C2's 5 922 dense windows: 79 of the first count's 270 us); `sub_heavy[j]`: which of the
    // group's sub-windows g + 16 j are heavy items, bit g
    bool shared_subs = false;
    uint32_t sub_heavy[2] = {0u, 0u};
    const int group_shift = (int)(threadIdx.x & 48u);   // the group's first lane in its wave
    // ranges of file `ff` for the window part [a, e) (a: first queried position, e: behind the last one)
    auto file_ranges = [&](int ff, int32_t tidx, int64_t a, int64_t e, int64_t s_lo, int64_t s_hi, FileRange &r) {
        const GFile g = ff == 0 ? gfile(file0) : gfile(files[ff]);
        const int64_t q0 = g.lin_off[tidx], qn = g.lin_off[tidx + 1] - q0 - 1;
        const int64_t e_up = e + (1 << kLinShift) - 1;   // (a table lookup rounds down: an end inside a bucket takes the whole bucket)
        if (e - a <= kExactSpan) {   // a short span: exact bounds
            r.lo = lin_exact<2>((const uint32_t PC_GLOBAL *)g.rec, g.lin_tab, q0, qn, a - Ws + 1);
            r.hi = lin_exact<2>((const uint32_t PC_GLOBAL *)g.rec, g.lin_tab, q0, qn, e);
            if (r.hi < r.lo) r.hi = r.lo;
            r.rlo = g.nrunrec ? (uint32_t)lin_exact<2>((const uint32_t PC_GLOBAL *)g.run_rec, g.rlin_tab, q0, qn, a - Wr + 1) : 0u;
            r.rhi = g.nrunrec ? (uint32_t)lin_exact<2>((const uint32_t PC_GLOBAL *)g.run_rec, g.rlin_tab, q0, qn, e) : 0u;
            if (r.rhi < r.rlo) r.rhi = r.rlo;
        } else {
            r.lo = lin_floor(g.lin_tab, q0, qn, a - Ws + 1);
            r.hi = lin_floor(g.lin_tab, q0, qn, e_up);
            r.rlo = g.nrunrec ? (uint32_t)lin_floor(g.rlin_tab, q0, qn, a - Wr + 1) : 0u;
            r.rhi = g.nrunrec ? (uint32_t)lin_floor(g.rlin_tab, q0, qn, e_up) : 0u;
        }
        r.glo = g.ngap ? lin_floor(g.glin_tab, q0, qn, a - W + 1) : 0;
        r.ghi = g.ngap ? lin_floor(g.glin_tab, q0, qn, e_up) : 0;
        r.llo = r.lhi = 0;
        if (g.nxlong) {   // long-span candidates of the whole queried span (every sub-window checks them)
            r.lhi = lin_floor(g.xllin_tab, q0, qn, s_hi);
            r.llo = lin_floor(g.xplin_tab, q0, qn, s_lo);
            if (r.llo > r.lhi) r.llo = r.lhi;
        }
    };
    // slices of a merged window for the ranges `r` of one file
    auto ceil_r = [&](int64_t n) -> uint32_t { return n > 0 ? (uint32_t)((n + R - 1) / R) : 0u; };
    auto merged_count = [&](const FileRange &r, bool first_file) -> uint32_t {
        if (b16) {
            const uint32_t c = ceil_r(r.hi - r.lo) + ceil_r((int64_t)r.rhi - (int64_t)r.rlo) + ceil_r(r.ghi - r.glo) + ceil_r(r.lhi - r.llo);
            return c ? c : (first_file ? 1u : 0u);
        }
        const int64_t nf = r.hi - r.lo;
        return nf > 0 ? ceil_r(nf) : ((r.ghi > r.glo || r.lhi > r.llo || r.rhi > r.rlo || first_file) ? 1u : 0u);
    };
    // (w: the window's fields filled in; il: light slots taken so far by this thread)
    auto merged_emit = [&](WorkItem &w, const FileRange &r, uint32_t base_l, uint32_t &il, bool first_file) {
        auto put = [&]() { const uint32_t slot = work_cap - 1u - (base_l + il++); if (slot < work_cap) work[slot] = w; };
        auto clear = [&]() { w.lo = w.hi = 0; w.glo = w.ghi = 0; w.llo = w.lhi = 0; w.rlo = w.rhi = 0u; w.sub_lo = 0; w.sub_hi = G; };
        if (b16) {   // one kind of range per slice
            uint32_t made = 0;
            for (int64_t a = r.lo; a < r.hi; a += R, ++made) { clear(); w.lo = a; w.hi = a + R < r.hi ? a + R : r.hi; put(); }
            for (int64_t a = r.rlo; a < (int64_t)r.rhi; a += R, ++made) { clear(); w.rlo = (uint32_t)a; w.rhi = (uint32_t)(a + R < (int64_t)r.rhi ? a + R : (int64_t)r.rhi); put(); }
            for (int64_t a = r.glo; a < r.ghi; a += R, ++made) { clear(); w.glo = a; w.ghi = a + R < r.ghi ? a + R : r.ghi; put(); }
            for (int64_t a = r.llo; a < r.lhi; a += R, ++made) { clear(); w.llo = a; w.lhi = a + R < r.lhi ? a + R : r.lhi; put(); }
            if (!made && first_file) { clear(); put(); }
            return;
        }
        const uint32_t cnt = merged_count(r, first_file);
        for (uint32_t k = 0; k < cnt; ++k) {
            clear();
            w.lo = r.lo + (int64_t)k * R;
            w.hi = (w.lo + R < r.hi) ? w.lo + R : r.hi;
            if (r.hi <= r.lo) { w.lo = r.lo; w.hi = r.hi; }
            if (k == 0) { w.glo = r.glo; w.ghi = r.ghi; w.llo = r.llo; w.lhi = r.lhi; w.rlo = r.rlo; w.rhi = r.rhi; }
            put();
        }
    };
    auto adds_of = [&](const FileRange &r) -> int64_t { return (r.hi - r.lo) + ((int64_t)r.rhi - (int64_t)r.rlo) + (r.ghi - r.glo) + (r.lhi - r.llo); };
    int64_t jn = 0;   // joint: records of all files in the window
    if (live && joint) {
        t = (int)idx;
        tl = tiles[t];
        ws = tl.win_start;
        const int64_t s_lo = ws + tl.span_lo, s_hi = ws + tl.span_hi + (1 << kLinShift) - 1, s_end = ws + tl.span_hi;
        int64_t ng = 0, nl = 0, nr = 0;
        for (int ff = 0; ff < nfiles; ++ff) {
            FileRange r;
            file_ranges(ff, tl.tid, s_lo, s_end, s_lo, s_hi, r);
            jn += r.hi - r.lo; ng += r.ghi - r.glo; nl += r.lhi - r.llo; nr += (int64_t)r.rhi - (int64_t)r.rlo;
        }
        const int64_t jt = jn + ng + nl + nr;   // everything the window's item would add
        while (S < kMaxSub && G % (S * 2 << kLinShift) == 0 && (b16 ? jt : jn) > R * S) S <<= 1;
        if (S == 1) {
            const bool small = small_g > 0 && (int)tl.span_hi - (int)tl.span_lo <= small_g && jn <= small_n && ng <= small_n &&
                               nl <= small_n && nr <= small_n;
            if (b16 && jt > kMax16) merge = true;
            else if (small) n_small = 1; else if (jn > R) n_heavy = 1; else n_light = 1;
        } else {
            const int sub = G / S;
            for (int k = 0; k < S; ++k) {
                const int64_t a = ws + (int64_t)k * sub;
                int64_t nk = 0, nk16 = 0;
                for (int ff = 0; ff < nfiles; ++ff) {   // (the very ranges the items below get: the class counts must agree with them)
                    FileRange r;
                    file_ranges(ff, tl.tid, a, a + sub, s_lo, s_hi, r);
                    nk += r.hi - r.lo;
                    nk16 += adds_of(r);
                }
                if (nk > pile || (b16 && nk16 > kMax16)) merge = true; // a pile-up inside one sub-window
                if (nk > R) ++n_heavy; else ++n_light;
            }
        }
        if (merge) {   // record slices of every file, merged through the compact histogram
            S = 1;
            n_heavy = 0;
            n_light = 0;
            for (int ff = 0; ff < nfiles; ++ff) {
                FileRange r;
                file_ranges(ff, tl.tid, s_lo, s_end, s_lo, s_hi, r);
                n_light += merged_count(r, ff == 0);
            }
        }
    }
    if (live && !joint) {
        t = (int)(idx / nfiles);
        f = (int)(idx % nfiles);
        tl = tiles[t];
        fv = f == 0 ? gfile(file0) : gfile(files[f]); // the first file's view travels as a kernel argument
        ws = tl.win_start;
        l0 = fv.lin_off[tl.tid];
        nb = fv.lin_off[tl.tid + 1] - l0 - 1;
        // only reads that can land on a queried position matter: a sparse annotation (one
        // 150-nt exon in a 4096-nt window) scans the exon's neighbourhood, not the whole window
        const int64_t s_lo = ws + tl.span_lo, s_hi = ws + tl.span_hi + (1 << kLinShift) - 1, s_end = ws + tl.span_hi;
        const bool exact = (int)tl.span_hi - (int)tl.span_lo <= kExactSpan;   // a short span: exact bounds instead of bucket edges
        if (exact) {   // both streams' bounds in one go
            const uint32_t PC_GLOBAL *const vv[4] = {(const uint32_t PC_GLOBAL *)fv.rec, (const uint32_t PC_GLOBAL *)fv.rec,
                                                     (const uint32_t PC_GLOBAL *)fv.run_rec, (const uint32_t PC_GLOBAL *)fv.run_rec};
            const uint32_t PC_GLOBAL *const ll[4] = {fv.lin_tab, fv.lin_tab, fv.rlin_tab, fv.rlin_tab};
            const int64_t kk[4] = {s_lo - Ws + 1, s_end, s_lo - Wr + 1, s_end};
            int64_t bound[4];
            lin_exact_multi<CG>(vv, ll, l0, nb, kk, fv.nrunrec ? 4 : 2, gsub, bound);
            wlo = bound[0]; whi = bound[1];
            if (whi < wlo) whi = wlo;
            if (fv.nrunrec) { wrlo = bound[2]; wrhi = bound[3]; if (wrhi < wrlo) wrhi = wrlo; }
        } else {
            wlo = lin_floor(fv.lin_tab, l0, nb, s_lo - Ws + 1);
            whi = lin_floor(fv.lin_tab, l0, nb, s_hi);
        }
        if (fv.ngap) {
            wglo = lin_floor(fv.glin_tab, l0, nb, s_lo - W + 1);
            wghi = lin_floor(fv.glin_tab, l0, nb, s_hi);
        }
        if (fv.nrunrec && !exact) { // aligned runs (of gapped and spliced reads) that start up to Wr before the span
            wrlo = lin_floor(fv.rlin_tab, l0, nb, s_lo - Wr + 1);
            wrhi = lin_floor(fv.rlin_tab, l0, nb, s_hi);
        }
        if (fv.nxlong) {
            // long-span reads outside the run stream that can reach the queried span: they start before
            // its end, and the running maximum of the ends (monotone) has passed its start
            // (two table lookups, both rounded outwards to a 128-nt bucket)
            lhi = lin_floor(fv.xllin_tab, l0, nb, s_hi);
            llo = lin_floor(fv.xplin_tab, l0, nb, s_lo);
            if (llo > lhi) llo = lhi;
        }
        const int64_t n = whi - wlo;
        const int64_t nt = n + (wghi - wglo) + (lhi - llo) + (wrhi - wrlo);   // everything the window's item would add
        merge = nfiles > 1;
        if (!merge) {
            while (S < kMaxSub && G % (S * 2 << kLinShift) == 0 && (b16 ? nt : n) > R * S) S <<= 1; // sub-windows end on index buckets
            if (S == 1) {
                const bool small = small_g > 0 && (int)tl.span_hi - (int)tl.span_lo <= small_g && n <= small_n &&
                                   (wghi - wglo) <= small_n && (lhi - llo) <= small_n && (wrhi - wrlo) <= small_n;
                if (b16 && nt > kMax16) merge = true;
                else if (small) n_small = 1; else if (n > R) n_heavy = 1; else n_light = 1;
            } else {
                const int sub = G / S;
                // what sub-window k scans: `mrg` -- it has to be merged; returns whether it is a heavy item
                auto sub_class = [&](int k, bool &mrg) -> bool {
                    const int64_t a = ws + (int64_t)k * sub;
                    // (the very range the item below gets -- exact lower bound for short sub-windows: the class counts must agree with it)
                    const int64_t lo_k = sub <= kExactSpan ? lin_exact<2>((const uint32_t PC_GLOBAL *)fv.rec, fv.lin_tab, l0, nb, a - Ws + 1)
                                                           : lin_floor(fv.lin_tab, l0, nb, a - Ws + 1);
                    int64_t nk = lin_floor(fv.lin_tab, l0, nb, a + sub) - lo_k;
                    if (nk < 0) nk = 0;
                    if (nk > pile) mrg = true; // a pile-up inside one sub-window
                    if (b16) {   // ... or a sub-window that would add more than a 16-bit bin holds (its runs and list entries counted in, by their widest bounds)
                        const int64_t gk = fv.ngap ? lin_floor(fv.glin_tab, l0, nb, a + sub) - lin_floor(fv.glin_tab, l0, nb, a - W + 1) : 0;
                        const int64_t rk = fv.nrunrec ? lin_floor(fv.rlin_tab, l0, nb, a + sub) - lin_floor(fv.rlin_tab, l0, nb, a - Wr + 1) : 0;
                        if (nk + gk + rk + (lhi - llo) > kMax16) mrg = true;
                    }
                    return nk > R;
                };
                if (CG == 16) {
                    bool mrg = false, hv[2] = {false, false};
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (gsub + 16 * j < S) hv[j] = sub_class(gsub + 16 * j, mrg);
                    // (the sixteen lanes of a group are here together: same window, same S)
                    merge = ((__ballot(mrg) >> group_shift) & 0xffffull) != 0ull;
                    sub_heavy[0] = (uint32_t)((__ballot(hv[0]) >> group_shift) & 0xffffull);
                    sub_heavy[1] = (uint32_t)((__ballot(hv[1]) >> group_shift) & 0xffffull);
                    n_heavy = (uint32_t)__popc(sub_heavy[0]) + (uint32_t)__popc(sub_heavy[1]);
                    n_light = (uint32_t)S - n_heavy;
                    shared_subs = !merge;
                } else {
                    for (int k = 0; k < S; ++k) {
                        if (sub_class(k, merge)) ++n_heavy; else ++n_light;
                    }
                }
            }
        }
        if (merge) {
            S = 1;
            n_heavy = 0;
            FileRange r;
            r.lo = wlo; r.hi = whi; r.glo = wglo; r.ghi = wghi; r.llo = llo; r.lhi = lhi; r.rlo = (uint32_t)wrlo; r.rhi = (uint32_t)wrhi;
            n_light = merged_count(r, f == 0);
        }
    }
    if (CG > 1 && gsub != 0) { n_heavy = 0; n_light = 0; n_small = 0; }   // (lane 0 of the group queues and writes)
    if (diag && live && gsub == 0 && f == 0 && lhi > llo) atomicAdd(&nwork[3], (uint32_t)(lhi - llo)); // PC_DEBUG_WORK: long-span candidates
    // one scan for the three classes: 21 bits each (a block queues far fewer than 2 M items)
    unsigned long long tot3;
    const unsigned long long off3 = block_scan_excl64((unsigned long long)n_heavy | ((unsigned long long)n_light << 21) |
                                                      ((unsigned long long)n_small << 42), s_wave64, tot3);
    const uint32_t m21 = (1u << 21) - 1u;
    const uint32_t tot_h = (uint32_t)tot3 & m21, tot_l = (uint32_t)(tot3 >> 21) & m21, tot_s = (uint32_t)(tot3 >> 42) & m21;
    const uint32_t off_h = (uint32_t)off3 & m21, off_l = (uint32_t)(off3 >> 21) & m21, off_s = (uint32_t)(off3 >> 42) & m21;
    if (threadIdx.x == 0) {
        s_base[0] = tot_h ? atomicAdd(&nwork[0], tot_h) : 0u;
        s_base[1] = tot_l ? atomicAdd(&nwork[1], tot_l) : 0u;
        s_base[2] = tot_s ? atomicAdd(&nwork[2], tot_s) : 0u;
    }
    __syncthreads();
    if (!(n_heavy + n_light + n_small) && !shared_subs) return;
    if (joint) {
        WorkItem w;
        w.tile = (uint32_t)t;
        w.file = 0u;
        w.mode_mask = tl.mode_mask;
        w.piece_begin = tl.piece_begin; w.piece_end = tl.piece_end;
        w.op_begin = tl.op_begin; w.op_end = tl.op_end;
        w.win_start = tl.win_start;
        w.span_lo = tl.span_lo; w.span_hi = tl.span_hi;
        w.merge = merge ? 1u : 0u;
        const int64_t s_lo = ws + tl.span_lo, s_hi = ws + tl.span_hi + (1 << kLinShift) - 1, s_end = ws + tl.span_hi;
        auto head = [&](const FileRange &r) {
            w.lo = r.lo; w.hi = r.hi; w.glo = r.glo; w.ghi = r.ghi; w.llo = r.llo; w.lhi = r.lhi; w.rlo = r.rlo; w.rhi = r.rhi;
        };
        if (n_small) {
            const uint32_t slot = s_base[2] + off_s;
            for (int ff = 0; ff < nfiles; ++ff) {
                FileRange r;
                file_ranges(ff, tl.tid, s_lo, s_end, s_lo, s_hi, r);
                if (ff == 0) head(r); else chain_small[(size_t)slot * (size_t)(nfiles - 1) + (size_t)(ff - 1)] = r;
            }
            w.win_start = tl.win_start + (int32_t)tl.span_lo; // a small window that starts at the first queried position
            w.sub_lo = 0; w.sub_hi = small_g;
            w.span_lo = 0; w.span_hi = (uint16_t)(tl.span_hi - tl.span_lo);
            work_small[slot] = w;
            return;
        }
        uint32_t ih = s_base[0] + off_h, il = s_base[1] + off_l;
        if (merge) {
            atomicAdd(&tile_items[t], n_light); // > 0 marks the tile for k_gather_split
            atomicAdd(&nwork[4], 1u);           // windows merged through the compact histogram (none: k_gather_split has nothing to do)
            uint32_t made = 0;
            for (int ff = 0; ff < nfiles; ++ff) {
                FileRange r;
                file_ranges(ff, tl.tid, s_lo, s_end, s_lo, s_hi, r);
                w.file = (uint32_t)ff;
                merged_emit(w, r, il, made, ff == 0);
            }
            return;
        }
        const int sub = G / S;
        for (int k = 0; k < S; ++k) {
            const int64_t a = S == 1 ? s_lo : ws + (int64_t)k * sub;
            const int64_t e = S == 1 ? s_end : a + sub;
            FileRange r0, r;
            file_ranges(0, tl.tid, a, e, s_lo, s_hi, r0);
            int64_t nk = r0.hi - r0.lo;
            for (int ff = 1; ff < nfiles; ++ff) {   // (counted first: the class decides the slot the chain entries belong to)
                file_ranges(ff, tl.tid, a, e, s_lo, s_hi, r);
                nk += r.hi - r.lo;
            }
            const uint32_t slot = nk > R ? ih++ : work_cap - 1u - (il++);
            if (slot >= work_cap) continue; // capacity is an upper bound; the test is defensive
            for (int ff = 1; ff < nfiles; ++ff) {
                file_ranges(ff, tl.tid, a, e, s_lo, s_hi, r);
                chain[(size_t)slot * (size_t)(nfiles - 1) + (size_t)(ff - 1)] = r;
            }
            head(r0);
            w.sub_lo = S == 1 ? 0 : k * sub;
            w.sub_hi = S == 1 ? G : w.sub_lo + sub;
            work[slot] = w;
        }
        return;
    }
    WorkItem w;
    w.tile = (uint32_t)t;
    w.file = (uint32_t)f;
    w.mode_mask = tl.mode_mask;
    w.piece_begin = tl.piece_begin; w.piece_end = tl.piece_end;
    w.op_begin = tl.op_begin; w.op_end = tl.op_end;
    w.win_start = tl.win_start;
    w.span_lo = tl.span_lo; w.span_hi = tl.span_hi;
    w.merge = merge ? 1u : 0u;
    if (n_small) {
        w.lo = wlo; w.hi = whi; w.glo = wglo; w.ghi = wghi; w.llo = llo; w.lhi = lhi;
        w.rlo = (uint32_t)wrlo; w.rhi = (uint32_t)wrhi;
        w.win_start = tl.win_start + (int32_t)tl.span_lo; // a small window that starts at the first queried position
        w.sub_lo = 0; w.sub_hi = small_g;
        w.span_lo = 0; w.span_hi = (uint16_t)(tl.span_hi - tl.span_lo);
        work_small[s_base[2] + off_s] = w;
        return;
    }
    uint32_t ih = s_base[0] + off_h, il = s_base[1] + off_l;
    if (merge) {
        atomicAdd(&tile_items[t], n_light); // > 0 marks the tile for k_gather_split
        atomicAdd(&nwork[4], 1u);
        FileRange r;
        r.lo = wlo; r.hi = whi; r.glo = wglo; r.ghi = wghi; r.llo = llo; r.lhi = lhi; r.rlo = (uint32_t)wrlo; r.rhi = (uint32_t)wrhi;
        uint32_t made = 0;
        merged_emit(w, r, il, made, f == 0);
        return;
    }
    const int sub = G / S;
    if (shared_subs) {   // the group's lanes write the items of their own sub-windows, in the slots the one-lane loop below would give them
        const uint32_t ih0 = s_base[0] + (uint32_t)__shfl((int)off_h, group_shift, 64), il0 = s_base[1] + (uint32_t)__shfl((int)off_l, group_shift, 64);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = gsub + 16 * j;
            if (k >= S) continue;
            const int64_t a = ws + (int64_t)k * sub, e = a + sub;
            if (sub <= kExactSpan) {
                w.lo = lin_exact<2>((const uint32_t PC_GLOBAL *)fv.rec, fv.lin_tab, l0, nb, a - Ws + 1);
                w.hi = lin_floor(fv.lin_tab, l0, nb, e);
                w.rlo = fv.nrunrec ? (uint32_t)lin_exact<2>((const uint32_t PC_GLOBAL *)fv.run_rec, fv.rlin_tab, l0, nb, a - Wr + 1) : 0u;
                w.rhi = fv.nrunrec ? (uint32_t)lin_floor(fv.rlin_tab, l0, nb, e) : 0u;
                if (w.hi < w.lo) w.hi = w.lo;
                if (w.rhi < w.rlo) w.rhi = w.rlo;
            } else {
                w.lo = lin_floor(fv.lin_tab, l0, nb, a - Ws + 1);
                w.hi = lin_floor(fv.lin_tab, l0, nb, e);
                w.rlo = fv.nrunrec ? (uint32_t)lin_floor(fv.rlin_tab, l0, nb, a - Wr + 1) : 0u;
                w.rhi = fv.nrunrec ? (uint32_t)lin_floor(fv.rlin_tab, l0, nb, e) : 0u;
            }
            w.glo = fv.ngap ? lin_floor(fv.glin_tab, l0, nb, a - W + 1) : 0;
            w.ghi = fv.ngap ? lin_floor(fv.glin_tab, l0, nb, e) : 0;
            w.llo = llo; w.lhi = lhi;
            w.sub_lo = k * sub;
            w.sub_hi = w.sub_lo + sub;
            const uint32_t below = (1u << gsub) - 1u;
            const uint32_t heavy_before = j == 0 ? (uint32_t)__popc(sub_heavy[0] & below) : (uint32_t)__popc(sub_heavy[0]) + (uint32_t)__popc(sub_heavy[1] & below);
            const bool heavy = ((sub_heavy[j] >> gsub) & 1u) != 0u;
            const uint32_t slot = heavy ? ih0 + heavy_before : work_cap - 1u - (il0 + ((uint32_t)k - heavy_before));
            if (slot < work_cap) work[slot] = w; // capacity is an upper bound; the test is defensive
        }
        return;
    }
    for (int k = 0; k < S; ++k) {
        const int64_t a = S == 1 ? ws + tl.span_lo : ws + (int64_t)k * sub;
        const int64_t e = S == 1 ? ws + tl.span_hi + (1 << kLinShift) - 1 : a + sub;
        if (S == 1) {   // (the window's own bounds from above: exact for a short span)
            w.lo = wlo; w.hi = whi; w.rlo = (uint32_t)wrlo; w.rhi = (uint32_t)wrhi;
        } else if (sub <= kExactSpan) {   // (a sub-window ends on a bucket edge: only its lower bounds need the search)
            w.lo = lin_exact<2>((const uint32_t PC_GLOBAL *)fv.rec, fv.lin_tab, l0, nb, a - Ws + 1);
            w.hi = lin_floor(fv.lin_tab, l0, nb, e);
            w.rlo = fv.nrunrec ? (uint32_t)lin_exact<2>((const uint32_t PC_GLOBAL *)fv.run_rec, fv.rlin_tab, l0, nb, a - Wr + 1) : 0u;
            w.rhi = fv.nrunrec ? (uint32_t)lin_floor(fv.rlin_tab, l0, nb, e) : 0u;
            if (w.hi < w.lo) w.hi = w.lo;
            if (w.rhi < w.rlo) w.rhi = w.rlo;
        } else {
            w.lo = lin_floor(fv.lin_tab, l0, nb, a - Ws + 1);
            w.hi = lin_floor(fv.lin_tab, l0, nb, e);
            w.rlo = fv.nrunrec ? (uint32_t)lin_floor(fv.rlin_tab, l0, nb, a - Wr + 1) : 0u;
            w.rhi = fv.nrunrec ? (uint32_t)lin_floor(fv.rlin_tab, l0, nb, e) : 0u;
        }
        w.glo = fv.ngap ? lin_floor(fv.glin_tab, l0, nb, a - W + 1) : 0;
        w.ghi = fv.ngap ? lin_floor(fv.glin_tab, l0, nb, e) : 0;
        w.llo = llo; w.lhi = lhi; // every sub-window checks the (few) long-span candidates
        w.sub_lo = S == 1 ? 0 : k * sub;
        w.sub_hi = S == 1 ? G : w.sub_lo + sub;
        const uint32_t slot = (w.hi - w.lo) > R ? ih++ : work_cap - 1u - (il++);
        if (slot < work_cap) work[slot] = w; // capacity is an upper bound; the test is defensive
    }
}

// ---------------------------------------------------------------- k_hist_point
// One workgroup per work item.  Streams its records once (coalesced 8-byte
// loads), bins every read's mapped position with LDS atomics into a window of G
// genome positions per strand mode, then writes the island pieces of the window
// to the compact histogram (plain coalesced stores when the tile has a single
// work item, global atomics otherwise).
// Per-work-item constants of the histogram kernel (all wave-uniform -> SGPRs).
struct HistCfg {
    int32_t win_start;
    uint32_t G;
    int base[kModes];      // LDS word offset of each strand mode's bins, -1 = mode absent
    uint32_t fmin, frange; // size filter as one unsigned compare: (L - fmin) <= frange
    int tab_lo;            // offset tables staged in LDS for aligned lengths [tab_lo, tab_lo + tab_n)
    uint32_t tab_n;
};

// Index from the left end of read.positions under the forward / reverse rule (-1: read not
// mapped) and the LDS row offset (stratified).  Offset tables come from LDS as one packed
// word per length: low half = forward_offsets[L], high half = reverse_offsets[L], 0xffff = none.
template <int KIND>
__device__ __forceinline__ void map_both(const MapParams &mp, const HistCfg &c, const uint32_t *ltab, int L, int &kf,
                                         int &kr, uint32_t &rowoff) {
    rowoff = 0;
    if (KIND == 0 || KIND == 1) { // :343-355 / :442-454
        const bool ok = mp.param < L;
        const int a = mp.param, z = L - 1 - mp.param;
        kf = ok ? (KIND == 0 ? a : z) : -1;
        kr = ok ? (KIND == 0 ? z : a) : -1;
    } else {
        const uint32_t t = (uint32_t)(L - c.tab_lo);
        uint32_t e = 0xffffffffu;
        const bool known = (uint32_t)L < (uint32_t)mp.table_len;
        if (t < c.tab_n) e = ltab[t];
        else if (known) { // rare: an aligned length outside the LDS-staged slice of the tables
            const int gf = ((const int32_t PC_GLOBAL *)mp.fw)[L], gr = ((const int32_t PC_GLOBAL *)mp.rc)[L];
            e = (uint32_t)(gf < 0 ? 0xffff : gf) | ((uint32_t)(gr < 0 ? 0xffff : gr) << 16);
        }
        const int f = (e & 0xffffu) == 0xffffu ? -1 : (int)(e & 0xffffu);
        const int r = (e >> 16) == 0xffffu ? -1 : (int)(e >> 16);
        if (KIND == 3) { // :625-638
            kf = known ? f : -1;
            kr = known ? r : -1;
        } else { // :765-778: no bad-offset check, read_positions[-1]
            const bool in = (L >= mp.min_len) & (L <= mp.max_len) & (L >= 1) & known;
            kf = in ? (f < 0 ? L - 1 : f) : -1;
            kr = in ? (r < 0 ? L - 1 : r) : -1;
            rowoff = (uint32_t)(L - mp.min_len) * c.G;
        }
    }
}

// Bin one read.  `pf`/`pr` = window-relative position under the forward / reverse rule.
// Modes 0 ('+') and 1 ('-') are mutually exclusive per read (strand filter), so they share
// one predicated ds_add; '.' (mode 2) and the unfiltered reverse rule (mode 3) add their own.
// B16: the bins are 16 bits wide, two positions per LDS word (the stratified rule: see k_hist_point) -- bin h is the
// (h & 1)-th half of word h >> 1, and one count is 1 << 16 (h & 1) added to that word.
template <bool B16>
__device__ __forceinline__ void bin_add(uint32_t *bins, uint32_t h) {
    if (B16) atomicAdd(&bins[h >> 1], 1u << ((h & 1u) << 4));
    else atomicAdd(&bins[h], 1u);
}

template <bool B16>
__device__ __forceinline__ void hist_bin(const HistCfg &c, bool valid, bool rev, int kf, int kr, uint32_t df,
                                         uint32_t dr, uint32_t rowoff, uint32_t *bins) {
    if ((c.base[0] & c.base[1]) != -1) { // uniform: the tile has a '+' and/or a '-' island
        const int k = rev ? kr : kf;
        const int b = rev ? c.base[1] : c.base[0];
        const uint32_t d = rev ? dr : df;
        if (valid & (k >= 0) & (b >= 0) & (d < c.G)) bin_add<B16>(bins, (uint32_t)b + rowoff + d);
    }
    if (c.base[2] >= 0) {
        if (valid & (kf >= 0) & (df < c.G)) bin_add<B16>(bins, (uint32_t)c.base[2] + rowoff + df);
    }
    if (c.base[3] >= 0) {
        if (valid & (kr >= 0) & (dr < c.G)) bin_add<B16>(bins, (uint32_t)c.base[3] + rowoff + dr);
    }
}

// ---- table-driven binning of the record stream.  Everything that depends only on (aligned
// length, strand mode) -- the index rule of the mapping function, the size filter, whether the
// tile has bins for the mode, the stratified row, the window origin -- is folded once per work
// item into a 4-byte LDS entry
//     ftab[L*4 + mode] = ((k - win_start) & 0xffff) << 16  |  LDS word offset of the mode's bins
// (an offset of 32768 - win_start marks "no bin for this pair": it lands outside every window).
// Adding the entry to the stream word puts the window-relative position in the high half (the
// low halves never carry: 0xff5 + 0x3fff < 2^16), so binning a stream record is: one ds_read_b32,
// add, shift, shift-or (the skip bit poisons the position), compare, and, add-shift, ds_add.
constexpr int kOpStage = 32;    // output pieces of a window staged in LDS ahead of the epilogue

template <int KIND>
__device__ __forceinline__ void fast_table_init(const MapParams &mp, const HistCfg &c, uint32_t mode_mask, int lo, int hi,
                                                uint32_t *ftab, uint32_t bins_word, int tid, int nthreads, int pre_f,
                                                int pre_r) {
    for (int i = tid; i < (hi - lo + 1) * kModes; i += nthreads) {
        const int L = lo + (i >> 2), m = i & 3;
        int row;
        // modes 1 and 3 use the reverse rule; the first round's table values were fetched at kernel start
        const int k = i == tid ? map_kleft_val<KIND>(mp, L, (m & 1) != 0, row, (m & 1) ? pre_r : pre_f)
                               : map_kleft<KIND>(mp, L, (m & 1) != 0, row);
        const bool have = (mode_mask >> m) & 1u;
        const int slot = __popc(mode_mask & ((1u << m) - 1u));
        const bool ok = have & (k >= 0) & ((uint32_t)(L - (int)c.fmin) <= c.frange);
        const uint32_t off = (uint32_t)((ok ? k : 32768) - c.win_start) & 0xffffu;
        const uint32_t base = ok ? bins_word + (uint32_t)(slot * mp.rows + row) * c.G : 0u;
        ftab[L * kModes + m] = (off << 16) | base;
    }
}

// Bin N records of the 4-byte stream.  All table reads of a strand-mode pass are issued before
// its first ds_add: the compiler cannot move an LDS read across an LDS atomic on its own (table and
// bins share the LDS), and one read-wait-add chain per record would serialise on the LDS latency.
// A record without a bin (outside the window, skipped, unmapped length) is sent to the lane's own
// dump word instead of being branched around.
template <int N, bool B16>
__device__ __forceinline__ void fast_bin(const uint32_t *ftab, uint32_t mode_mask, uint32_t G, uint32_t dump,
                                         const uint32_t (&w)[N], uint32_t *smem) {
    const char *tab = (const char *)ftab;
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        // pass 0: '+' / '-' (mutually exclusive per read: entry picked by the read's strand);
        // pass 1: '.' (all reads, forward rule); pass 2: all reads, reverse rule
        if (!(mode_mask & (pass == 0 ? 3u : (pass == 1 ? 4u : 8u)))) continue;
        uint32_t e[N];
#pragma unroll
        for (int i = 0; i < N; ++i)
            e[i] = *(const uint32_t *)(tab + (pass == 0 ? (w[i] & 0xff4u) : (w[i] & 0xff0u) + (pass == 1 ? 8u : 12u)));
        // byte address of every record's bin, then one ds_add per run of equal addresses:
        // neighbouring stream records are neighbours in the coordinate-sorted file, so the reads
        // piled on one position collapse into a single LDS atomic per lane
        uint32_t addr[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const uint32_t d = (w[i] + e[i]) >> 16;
            // (B16: the entry's low half is the bin's index in 16-bit units -- a byte address with 2-byte granularity)
            addr[i] = (((w[i] << 31) | d) < G) ? ((e[i] & 0xffffu) + d) << (B16 ? 1 : 2) : dump;
        }
        uint32_t cnt = 1;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const bool last = (i == N - 1) || (addr[i + 1 < N ? i + 1 : i] != addr[i]);
            if (last) {
                if (B16) atomicAdd((uint32_t *)((char *)smem + (addr[i] & ~3u)), cnt << ((addr[i] & 2u) << 3));   // (the dump word is word-aligned: low half)
                else atomicAdd((uint32_t *)((char *)smem + addr[i]), cnt);
            }
            cnt = last ? 1u : cnt + 1u;
        }
    }
}

// Bin one record of the run stream (one aligned run of a gapped / spliced read) through the entry table -- see the
// run-stream loop of k_hist_point.  A lane without a record holds an excluded one.
template <bool B16>
__device__ __forceinline__ void run_bin(const uint32_t *ftab, uint32_t mode_mask, uint32_t G, uint32_t dump, u32x2 rr,
                                        uint32_t win_start, uint32_t *smem) {
    const char *tab = (const char *)ftab;
    const uint32_t len = rr.y & 0xffu, cum = (rr.y >> 8) & 0xffu, fl = rr.y >> 24;
    const uint32_t lbyte = (rr.y >> 12) & 0xff0u;   // L * 16: byte offset of the length's four entries
    const bool live = (fl & kFlagExcluded) == 0u;
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        if (!(mode_mask & (pass == 0 ? 3u : (pass == 1 ? 4u : 8u)))) continue;
        const uint32_t e = *(const uint32_t *)(tab + lbyte + (pass == 0 ? (fl & kFlagReverse) << 2 : (pass == 1 ? 8u : 12u)));
        const uint32_t koff = (e >> 16) - cum;                       // (k - cum - win_start) mod 2^16 in the low half
        const uint32_t t = (koff + win_start) & 0xffffu;             // k - cum: inside this run when < len
        const uint32_t d = (koff + rr.x) & 0xffffu;                  // window-relative position of read.positions[k]
        const uint32_t addr = (live & (t < len) & (d < G)) ? ((e & 0xffffu) + d) << (B16 ? 1 : 2) : dump;
        if (B16) atomicAdd((uint32_t *)((char *)smem + (addr & ~3u)), 1u << ((addr & 2u) << 3));
        else atomicAdd((uint32_t *)((char *)smem + addr), 1u);
    }
}

// The record stream of one work item: 16-byte loads of four 4-byte records, every wave owns
// 4 KiB per batch (U x 64 lanes x 16 B, the U loads of a lane 1 KiB apart -> one address register
// and immediate offsets), register double buffer.  Batches that lie wholly inside the range are
// loaded without per-lane predicates; lanes past the end of the last batch hold skip words.
template <int WG, int U, bool B16>
__device__ __forceinline__ void stream_records(const u32x4 PC_GLOBAL *src, int nquads, u32x4 (&cur)[U], const u32x4 none,
                                               const uint32_t *ftab, uint32_t mode_mask, uint32_t G, uint32_t dump,
                                               uint32_t *smem) {
    const int lane_j = (int)(threadIdx.x >> 6) * (64 * U) + (int)(threadIdx.x & 63);
    // two register sets used alternately (the loop body is written out twice): handing the next
    // batch over with register moves would make every iteration wait for all of its loads
    auto load = [&](u32x4 (&d)[U], int nb) {
        const u32x4 PC_GLOBAL *q = src + nb + lane_j;
        if (nb + WG * U <= nquads) {
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = q[u * 64];
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = (nb + lane_j + u * 64 < nquads) ? q[u * 64] : none;
        }
    };
    auto bin = [&](const u32x4 (&c)[U], int base) {
        const int wave_q = base + (int)(threadIdx.x >> 6) * (64 * U); // first quad of this wave's slice of the batch
#pragma unroll
        for (int u = 0; u < U; u += 2) { // eight records (two quads of the lane's slice) per call
            if (wave_q + u * 64 >= nquads) break; // wave-uniform: nothing but padding from here on (sparse windows)
            const uint32_t w8[8] = {c[u].x, c[u].y, c[u].z, c[u].w, c[u + 1].x, c[u + 1].y, c[u + 1].z, c[u + 1].w};
            fast_bin<8, B16>(ftab, mode_mask, G, dump, w8, smem);
        }
    };
    u32x4 alt[U];
    for (int base = 0; base < nquads; base += 2 * WG * U) {
        load(alt, base + WG * U);
        bin(cur, base);
        if (base + WG * U >= nquads) break;
        load(cur, base + 2 * WG * U);
        bin(alt, base + WG * U);
    }
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

// One workgroup per work item.  The work-item descriptor carries the tile's window and
// piece range, and up to two staged files travel as kernel arguments, so the first record
// batch is requested after a single dependent load; the LDS set-up runs under that latency.
// OUTMODE: 0 = int64 counts, 1 = float64 counts, 2 = float64 reads-per-million
// (count / sum * 1e6 in that order, genome_array.py:826-827)
template <int OUTMODE> struct OutT_ { typedef int64_t type; };
template <> struct OutT_<1> { typedef double type; };
template <> struct OutT_<2> { typedef double type; };
template <int OUTMODE>
__device__ __forceinline__ typename OutT_<OUTMODE>::type out_conv(uint32_t v, double norm_sum) {
    if (OUTMODE == 0) return (typename OutT_<OUTMODE>::type)v;
    if (OUTMODE == 1) return (typename OutT_<OUTMODE>::type)(double)v;
    return (typename OutT_<OUTMODE>::type)((double)v / norm_sum * 1e6);
}

// out_step == 0: the slice is not laid out but SUMMED into one element per row (region
// statistics: numpy.nansum(chain.get_masked_counts(ga)), bin/counts_in_region.py:120).  Integer
// counts, so the atomic adds are exact and order-independent (int64, or float64 below 2^53).
template <int OUTMODE>
__device__ __forceinline__ void out_add(typename OutT_<OUTMODE>::type *dst, unsigned long long partial) {
    for (int o = 32; o > 0; o >>= 1) partial += __shfl_down(partial, o, 64);
    if ((threadIdx.x & 63) == 0 && partial) {
        if (OUTMODE == 0) atomicAdd((unsigned long long *)dst, partial);
        else atomicAdd((double *)dst, (double)partial);
    }
}

// MULTI: several alignment files (joint windows: the item's chain of per-file ranges is walked) -- an instantiation of
// its own, so that the single-file kernels keep their register budget (seven waves per SIMD for the variable rule)
// Experiment hook (scripts/exp_hist_sections.py; never set in the product build): a build with -DPC_HIST_SKIP=<mask>
// leaves sections of the kernel out -- 1 record stream, 2 side lists, 4 epilogue, 8 entry table, 16 bin clear -- so
// that their share of a launch can be timed (the counts are then wrong).
#ifndef PC_HIST_SKIP
#define PC_HIST_SKIP 0
#endif
// SINGLE: a plan of ONE window over one file (`ga[segment]`, the reference's scripts ask region by region,
// genome_array.py:861-928): the workgroup looks its record ranges up itself (what k_tile_ranges does for a window that is
// not cut: a handful of linear-index reads) -- the whole count is this one launch, no work list, no second class, no
// merge pass.  `work` then points at the plan's tile, `chain` carries the halos {Ws, Wg, Wr}.
template <int KIND, int OUTMODE, int WG, bool SMALL, bool MULTI, bool SINGLE = false>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(MULTI ? 5 : PC_HIST_WAVES(KIND), 8))) void k_hist_point(const Piece *__restrict__ pieces,
                                                    const OutPiece *__restrict__ opieces, FileView file0,
                                                    FileView file1, const FileView *__restrict__ files,
                                                    const WorkItem *__restrict__ work,
                                                    const uint32_t *__restrict__ nwork,
                                                    const uint32_t *__restrict__ tile_items, MapParams mp,
                                                    int G, int max_slots, int tab_lo, int tab_n, int fast_lo,
                                                    int fast_hi, uint32_t *hist,
                                                    int64_t hist_row_stride, typename OutT_<OUTMODE>::type *out,
                                                    double norm_sum, uint32_t work_cap, uint32_t grid_front,
                                                    const FileRange *__restrict__ chain, int nfiles,
                                                    // SINGLE with work == nullptr (pc_query_segment: `ga[segment]` in ONE call, no plan object, no upload, no
                                                    // read-back copy): the window and its one output piece travel in the kernel's arguments, `out` is page-locked
                                                    // host memory the kernel writes itself, and `done` -- behind the counts -- tells the polling host they are there
                                                    Tile q_tile, OutPiece q_op, uint32_t *done, uint32_t done_seq) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    // The stratified rule (the only one with several rows) bins into 16-BIT counters, two positions per LDS word: the
    // same bytes of LDS hold a window twice as long, so a plan has half the windows -- half the workgroup starts, entry
    // tables, bin clears and halos (C5: 798 k windows of 256 positions x 11 rows, 0.8 of 3.7 ms in their fixed cost).
    // What makes it safe: no work item of such a plan adds 65 536 times or more (k_tile_ranges cuts or merges the windows
    // whose records, runs and list entries could), so no bin carries into its neighbour.
    constexpr bool B16 = KIND == 4;
    // heavy items sit at the front of the list, light ones at the back (see k_tile_ranges);
    // the sparse-window list is a plain array of its own
    // The grid spans the whole list capacity and block b serves slot b (heavy slots [0, n_heavy),
    // light slots [cap - n_light, cap), nothing in between), so the item is requested together with
    // the counters instead of after them: one dependent round trip less per workgroup.
    // (the first `grid_front` blocks serve the front of the list, the others its back: the host launches
    // the whole capacity the first time and exactly the queued counts once it has seen them)
    const uint32_t slot = SINGLE ? 0u : (blockIdx.x < grid_front ? blockIdx.x : work_cap - (gridDim.x - blockIdx.x));
    const uint32_t n_heavy = SINGLE ? 1u : (SMALL ? nwork[2] : nwork[0]), n_light = (SMALL || SINGLE) ? 0u : nwork[1];
    WorkItem w_;
    if (SINGLE) {
        const Tile tl = work ? *(const Tile *)work : q_tile;
        const GFile g0 = gfile(file0);
        const int Ws = (int)hist_row_stride, Wg = (int)work_cap, Wr = (int)grid_front;   // (the halos travel in arguments a single window has no use for)
        const int64_t q0 = g0.lin_off[tl.tid], qn = g0.lin_off[tl.tid + 1] - q0 - 1;
        const int64_t a = (int64_t)tl.win_start + tl.span_lo, e = (int64_t)tl.win_start + tl.span_hi + (1 << kLinShift) - 1;
        w_.lo = lin_floor(g0.lin_tab, q0, qn, a - Ws + 1);
        w_.hi = lin_floor(g0.lin_tab, q0, qn, e);
        w_.glo = g0.ngap ? lin_floor(g0.glin_tab, q0, qn, a - Wg + 1) : 0;
        w_.ghi = g0.ngap ? lin_floor(g0.glin_tab, q0, qn, e) : 0;
        w_.rlo = g0.nrunrec ? (uint32_t)lin_floor(g0.rlin_tab, q0, qn, a - Wr + 1) : 0u;
        w_.rhi = g0.nrunrec ? (uint32_t)lin_floor(g0.rlin_tab, q0, qn, e) : 0u;
        w_.llo = w_.lhi = 0;
        if (g0.nxlong) {
            w_.lhi = lin_floor(g0.xllin_tab, q0, qn, e);
            w_.llo = lin_floor(g0.xplin_tab, q0, qn, a);
            if (w_.llo > w_.lhi) w_.llo = w_.lhi;
        }
        w_.tile = 0u; w_.file = 0u; w_.win_start = tl.win_start; w_.mode_mask = tl.mode_mask;
        w_.piece_begin = tl.piece_begin; w_.piece_end = tl.piece_end; w_.op_begin = tl.op_begin; w_.op_end = tl.op_end;
        w_.sub_lo = 0; w_.sub_hi = G; w_.merge = 0u; w_.span_lo = tl.span_lo; w_.span_hi = tl.span_hi;
    } else {
        w_ = work[slot];
    }
    const WorkItem w = w_;
    // offset-table values this thread needs for the LDS tables (variable / stratified rules): they
    // depend on kernel arguments only and travel together with the work item
    int pre_f = -1, pre_r = -1;
    if (KIND >= 3) {
        const int32_t PC_GLOBAL *fw = (const int32_t PC_GLOBAL *)mp.fw, *rc = (const int32_t PC_GLOBAL *)mp.rc;
        const int Lp = fast_lo + (int)(threadIdx.x >> 2);
        if (Lp <= fast_hi && Lp < mp.table_len) { pre_f = fw[Lp]; pre_r = rc[Lp]; }
    }
    if (!(slot < n_heavy || slot >= work_cap - n_light)) return;
    const GFile fv = w.file == 0 ? gfile(file0) : (w.file == 1 ? gfile(file1) : gfile(files[w.file]));

    // ---- first batch of the record stream (and the first gapped records).  The stream is read
    // as quads of 4-byte records; the records of the first quad that precede `lo` are masked
    // below, and a last quad that `hi` cuts is handled apart (by thread 0, masked on both sides).
    constexpr int U = PC_HIST_U(KIND);
    static_assert(U % 2 == 0, "PC_HIST_U must be even");
    const int64_t quad_lo = w.lo >> 2;
    const int nquads = (int)((w.hi >> 2) - quad_lo); // quads wholly below hi (may be 0, never negative)
    const u32x4 PC_GLOBAL *src = fv.stream4 + quad_lo;
    const u32x4 none = {kStreamSkip, kStreamSkip, kStreamSkip, kStreamSkip};
    u32x4 cur[U];
    {
        const int lane_j = (int)(threadIdx.x >> 6) * (64 * U) + (int)(threadIdx.x & 63);
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = (lane_j + u * 64 < nquads) ? src[lane_j + u * 64] : none;
    }
    u32x4 tail = none;
    if ((w.hi & 3) && threadIdx.x == 0) tail = fv.stream4[w.hi >> 2];
    // the window's output pieces (48 B each) are fetched now and parked in LDS, so that the
    // epilogue does not start with a chain of dependent global loads
    const int nstage = w.merge ? 0 : (int)min(w.op_end - w.op_begin, (uint32_t)kOpStage);
    u32x4 opq = {0u, 0u, 0u, 0u};
    if ((!SINGLE || work) && (int)threadIdx.x < nstage * 3) opq = ((const u32x4 PC_GLOBAL *)(opieces + w.op_begin))[threadIdx.x];
    const u32x4 gnone = {0u, kFlagExcluded << 16, 0u, 0u};
    // first batch of the gapped-record list, requested with everything else -- where the register budget is that of
    // six waves anyway (eight registers held across the stream loop are what separates six waves from seven)
    constexpr bool kGapPrefetch = MULTI || PC_HIST_WAVES(KIND) <= 6;
    const int64_t gj0 = w.glo + threadIdx.x;
    u32x4 gfirst = gnone;
    i32x4 gfirst_runs = {0, 1, 0, 0};
    if (kGapPrefetch && gj0 < w.ghi) { gfirst = fv.gap_rec[gj0]; gfirst_runs = fv.gap_runs[gj0]; }
    const u32x2 rnone = {0u, kFlagExcluded << 24};
    const uint32_t rj0 = w.rlo + threadIdx.x;
    const u32x2 rfirst = (rj0 < w.rhi) ? fv.run_rec[rj0] : rnone;

    HistCfg c;
    c.win_start = w.win_start;
    c.G = (uint32_t)G;
    int nslots = 0;
#pragma unroll
    for (int m = 0; m < kModes; ++m) c.base[m] = ((w.mode_mask >> m) & 1u) ? (nslots++) * mp.rows * G : -1;
    c.fmin = mp.filt_on ? (uint32_t)mp.filt_min : 0u;
    c.frange = (mp.filt_on && mp.filt_max != -1) ? (uint32_t)(mp.filt_max - mp.filt_min) : 0x7fffffffu - c.fmin;   // (no maximum: every length, wide reads included)
    c.tab_lo = tab_lo;
    c.tab_n = (uint32_t)tab_n;
    // LDS: [table-driven entries, indexed by aligned length from 0][packed offset tables][bins]
    const int fwords = (fast_hi + 1) * kModes;
    uint32_t *ftab = smem;
    uint32_t *ltab = smem + fwords;                      // variable / stratified rules: gapped and long-span reads
    uint32_t *bins = smem + fwords + ((tab_n + 3) & ~3);
    OutPiece *s_op = (OutPiece *)(bins + (((size_t)max_slots * mp.rows * G) >> (B16 ? 1 : 0))); // 16-byte aligned: every part is a multiple of 4 words
    const uint32_t dump = (uint32_t)((char *)(s_op + kOpStage) - (char *)smem) + (threadIdx.x & 63u) * 4u; // the lane's dump word
    if (SINGLE && !work) { if (threadIdx.x == 0) s_op[0] = q_op; }   // (the one output piece of an argument-borne window)
    else if ((int)threadIdx.x < nstage * 3) ((u32x4 *)s_op)[threadIdx.x] = opq;
    if (!(PC_HIST_SKIP & 16)) {   // only bins in [span_lo, span_hi) are ever read back: clear just those
        // (16-byte stores over the span rounded out to 4 words; no per-element division)
        // (a 16-byte store covers 4 bins of 32 bits, 8 of 16)
        constexpr int PS = B16 ? 3 : 2;
        const int lo4 = (int)w.span_lo >> PS, hi4 = ((int)w.span_hi + (1 << PS) - 1) >> PS, nrow = nslots * mp.rows;
        const u32x4 zero4 = {0u, 0u, 0u, 0u};
        if (nrow > 2 && (hi4 - lo4) * 2 > (G >> PS)) {
            // many rows and a span that covers most of the window: clear all rows in one flat sweep
            // (a row-by-row loop costs one mostly idle pass per row)
            u32x4 *all4 = (u32x4 *)bins;
            for (int i = (int)threadIdx.x; i < nrow * (G >> PS); i += WG) all4[i] = zero4;
        } else {
            for (int r = 0; r < nrow; ++r) {
                u32x4 *row4 = (u32x4 *)bins + r * (G >> PS);
                for (int i = lo4 + (int)threadIdx.x; i < hi4; i += WG) row4[i] = zero4;
            }
        }
    }
    // (the packed offset table serves the gapped-record and long-span lists only: the record stream and the run stream
    // go through the entry table -- a window without such records, the common case, does not build it)
    if (KIND >= 3 && (MULTI || w.ghi > w.glo || w.lhi > w.llo)) {
        const int32_t PC_GLOBAL *fw = (const int32_t PC_GLOBAL *)mp.fw, *rc = (const int32_t PC_GLOBAL *)mp.rc;
        for (int i = threadIdx.x; i < tab_n; i += WG) {
            const int f = fw[tab_lo + i], r = rc[tab_lo + i];
            ltab[i] = (uint32_t)(f < 0 ? 0xffff : f) | ((uint32_t)(r < 0 ? 0xffff : r) << 16);
        }
    }
    // (the entries carry the bins' place in units of one bin: words, or 16-bit halves)
    if (!(PC_HIST_SKIP & 8)) fast_table_init<KIND>(mp, c, w.mode_mask, fast_lo, fast_hi, ftab, (uint32_t)(bins - smem) << (B16 ? 1 : 0), (int)threadIdx.x, WG, pre_f, pre_r);
    if (threadIdx.x == 0) {
        // records of the first quad before `lo`, of the cut last quad outside [lo, hi)
        const int lead = (int)(w.lo & 3), keep = (int)(w.hi & 3);
        if (nquads > 0) {
            if (lead > 0) cur[0].x = kStreamSkip;
            if (lead > 1) cur[0].y = kStreamSkip;
            if (lead > 2) cur[0].z = kStreamSkip;
        }
        const int tlead = nquads == 0 ? lead : 0; // lo and hi inside the same quad
        if (keep <= 0 || tlead > 0) tail.x = kStreamSkip;
        if (keep <= 1 || tlead > 1) tail.y = kStreamSkip;
        if (keep <= 2 || tlead > 2) tail.z = kStreamSkip;
        tail.w = kStreamSkip;
    }
    __syncthreads();

    // ---- the record stream: no dependent global loads in this loop
    if (!(PC_HIST_SKIP & 1)) stream_records<WG, U, B16>(src, nquads, cur, none, ftab, w.mode_mask, c.G, dump, smem);
    else if (cur[0].x == 0x12345u) smem[0] = tail.x + opq.x + gfirst.x + rfirst.x + gfirst_runs.x;   // (keeps the loads of the prologue alive)
    if (!(PC_HIST_SKIP & 1) && (w.hi & 3) && threadIdx.x < 64) { // the cut last quad lives in lane 0 of the first wave
        const uint32_t w4[4] = {tail.x, tail.y, tail.z, tail.w};
        fast_bin<4, B16>(ftab, w.mode_mask, c.G, dump, w4, smem);
    }

    // the side lists of one file (run stream, gapped records outside it, long-span reads); `first_`: the head file of
    // the item, whose first batches were requested at kernel start
    auto side_lists = [&](const GFile &fv, uint32_t rlo_, uint32_t rhi_, int64_t glo_, int64_t ghi_, int64_t llo_, int64_t lhi_, bool first_) {
        // ---- run stream: the aligned runs of gapped and spliced reads (aligned length <= kStreamMaxLen), one
        // 8-byte record per run, sorted by run start.  A rule picks ONE index k of read.positions; the run
        // whose read indices [cum, cum + len) contain k holds the mapped position start + (k - cum), and that
        // run starts at most `len` before it -- so a window scans the runs that start up to Wr before its
        // first queried position: no dependent loads, no introns to look across, every run read once.
        // Binned through the same LDS entry table as the record stream: the entry of (L, strand mode) holds
        // (k - win_start) mod 2^16 and the bins' word offset, so  k - cum  (is the index inside this run?) and the
        // window-relative position  start + (k - cum) - win_start  are two adds each; size filter, rule, stratified
        // row and "no bins for this mode" are already folded into the entry (an entry without a bin holds k = 32768,
        // which lies in no run).
        // (four loads per lane in flight: a window's runs usually fit one trip, and a trip costs one memory latency)
        for (uint32_t base = rlo_; base < rhi_; base += 4 * WG) {
            u32x2 rr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t j = base + u * WG + threadIdx.x;
                rr[u] = (u == 0 && first_ && base == rlo_) ? rfirst : (j < rhi_ ? fv.run_rec[j] : rnone);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (base + u * WG < rhi_) run_bin<B16>(ftab, w.mode_mask, c.G, dump, rr[u], (uint32_t)c.win_start, smem);
        }

        // ---- gapped records outside the run stream (aligned length > kStreamMaxLen): their aligned runs live in a side
        // array; consecutive list entries own consecutive runs, so these gathers stay coalesced.
        for (int64_t base = glo_; base < ghi_; base += WG) {
            const int64_t j = base + threadIdx.x;
            const bool in = j < ghi_;
            const u32x4 g = (kGapPrefetch && first_ && base == glo_) ? gfirst : (in ? fv.gap_rec[j] : gnone);
            const i32x4 gr = (kGapPrefetch && first_ && base == glo_) ? gfirst_runs : (in ? fv.gap_runs[j] : i32x4{0, 1, 0, 0});
            const uint32_t meta = g.y, hi = meta >> 16;
            const int L = (int)(meta & 0xffffu), nb = (int)(meta >> 24);
            const bool valid = in & ((hi & kFlagExcluded) == 0) & ((uint32_t)(L - (int)c.fmin) <= c.frange);
            const i32x2 b0 = {gr.x, gr.y}, b1 = {gr.z, gr.w}; // first two runs travel with the list entry
            int kf, kr;
            uint32_t rowoff;
            map_both<KIND>(mp, c, ltab, L, kf, kr, rowoff);
            const int32_t pf = (valid && kf >= 0) ? walk_from(fv, g.z, nb, kf, b0, b1) : 0;
            const int32_t pr = (valid && kr >= 0) ? walk_from(fv, g.z, nb, kr, b0, b1) : 0;
            hist_bin<B16>(c, valid, hi & kFlagReverse, kf, kr, (uint32_t)(pf - c.win_start), (uint32_t)(pr - c.win_start), rowoff, bins);
        }

        // ---- long-span (spliced) reads that can reach this window: same binning, every run walked
        for (int64_t base = llo_; base < lhi_; base += WG) {
            const int64_t j = base + threadIdx.x;
            const bool in = j < lhi_;
            const u32x4 g = in ? fv.xlong_rec[j] : gnone;
            const i32x4 gr = in ? fv.xlong_runs[j] : i32x4{0, 1, 0, 0};
            const uint32_t meta = g.y, hi = meta >> 16;
            int L = (int)(meta & 0xffffu), nb = (int)(meta >> 24);
            if (hi & kFlagWide) { const u32x2 tv = fv.xlong_wide[j]; L = (int)tv.x; nb = (int)tv.y; }   // beyond the 16 / 8-bit fields
            const bool valid = in & ((hi & kFlagExcluded) == 0) & ((uint32_t)(L - (int)c.fmin) <= c.frange);
            const i32x2 b0 = {gr.x, gr.y}, b1 = {gr.z, gr.w};
            int kf, kr;
            uint32_t rowoff;
            map_both<KIND>(mp, c, ltab, L, kf, kr, rowoff);
            const int32_t pf = (valid && kf >= 0) ? walk_from(fv, g.z, nb, kf, b0, b1) : 0;
            const int32_t pr = (valid && kr >= 0) ? walk_from(fv, g.z, nb, kr, b0, b1) : 0;
            hist_bin<B16>(c, valid, hi & kFlagReverse, kf, kr, (uint32_t)(pf - c.win_start), (uint32_t)(pr - c.win_start), rowoff, bins);
        }
    };
    if (!(PC_HIST_SKIP & 2)) side_lists(fv, w.rlo, w.rhi, w.glo, w.ghi, w.llo, w.lhi, true);
    // ---- joint window (several alignment files): the records of the other files into the same bins
    if (MULTI && !w.merge) {
        for (int ff = 1; ff < nfiles; ++ff) {
            const FileRange fr = chain[(size_t)slot * (size_t)(nfiles - 1) + (size_t)(ff - 1)];
            const GFile fv2 = ff == 1 ? gfile(file1) : gfile(files[ff]);
            const int64_t ql2 = fr.lo >> 2;
            const int nq2 = (int)((fr.hi >> 2) - ql2);
            const u32x4 PC_GLOBAL *src2 = fv2.stream4 + ql2;
            u32x4 c2[U];
            {
                const int lane_j = (int)(threadIdx.x >> 6) * (64 * U) + (int)(threadIdx.x & 63);
#pragma unroll
                for (int u = 0; u < U; ++u) c2[u] = (lane_j + u * 64 < nq2) ? src2[lane_j + u * 64] : none;
            }
            u32x4 tail2 = none;
            if ((fr.hi & 3) && threadIdx.x == 0) tail2 = fv2.stream4[fr.hi >> 2];
            if (threadIdx.x == 0) {   // records of the first quad before `lo`, of the cut last quad outside [lo, hi)
                const int lead = (int)(fr.lo & 3), keep = (int)(fr.hi & 3);
                if (nq2 > 0) {
                    if (lead > 0) c2[0].x = kStreamSkip;
                    if (lead > 1) c2[0].y = kStreamSkip;
                    if (lead > 2) c2[0].z = kStreamSkip;
                }
                const int tlead = nq2 == 0 ? lead : 0;
                if (keep <= 0 || tlead > 0) tail2.x = kStreamSkip;
                if (keep <= 1 || tlead > 1) tail2.y = kStreamSkip;
                if (keep <= 2 || tlead > 2) tail2.z = kStreamSkip;
                tail2.w = kStreamSkip;
            }
            stream_records<WG, U, B16>(src2, nq2, c2, none, ftab, w.mode_mask, c.G, dump, smem);
            if ((fr.hi & 3) && threadIdx.x < 64) {
                const uint32_t w4[4] = {tail2.x, tail2.y, tail2.z, tail2.w};
                fast_bin<4, B16>(ftab, w.mode_mask, c.G, dump, w4, smem);
            }
            side_lists(fv2, fr.rlo, fr.rhi, fr.glo, fr.ghi, fr.llo, fr.lhi, false);
        }
    }
    __syncthreads();

    if (PC_HIST_SKIP & 4) return;
    if (!w.merge) {
        // ---- this workgroup owns [sub_lo, sub_hi) of the window and its bins are complete:
        // write every queried segment slice straight into the caller's layout (chain offset,
        // 5'->3' reversal, int64/float64, normalisation) -- SegmentChain.get_counts,
        // roitools.pyx:3259-3271
        // several rows: the slices are short (a 150-nt exon against 256 lanes) -- every wave takes slices of its own
        // (C5: 3.45 -> 3.39 ms)
        const bool per_wave = B16 && WG > 64;
        const uint32_t op_first = per_wave ? w.op_begin + (threadIdx.x >> 6) : w.op_begin, op_step = per_wave ? WG / 64 : 1;
        const int lane0 = per_wave ? (int)(threadIdx.x & 63) : (int)threadIdx.x, lanes = per_wave ? 64 : WG;
        for (uint32_t oi = op_first; oi < w.op_end; oi += op_step) {
            const uint32_t k = oi - w.op_begin;
            const OutPiece o = k < (uint32_t)kOpStage ? s_op[k] : opieces[oi];
            const int rel = o.start - w.win_start;
            const int i0 = w.sub_lo > rel ? w.sub_lo - rel : 0;
            const int i1 = (w.sub_hi - rel) < o.len ? (w.sub_hi - rel) : o.len;
            const int slot = __popc(w.mode_mask & ((1u << o.mode) - 1u));
            if (mp.rows > 1 && o.step != 0) {
                // several rows (stratified rule): a lane keeps its position and walks down the rows with two
                // pointer increments per element -- a row-major loop pays the 64-bit address set-up of a row
                // for 1.2 KB of output (a 150-nt exon), eleven times per piece
                for (int i = i0 + lane0; i < i1; i += lanes) {
                    typename OutT_<OUTMODE>::type *dstp = out + o.out_off + (int64_t)o.step * i;
                    if (B16) {
                        const uint16_t *srcp = (const uint16_t *)bins + slot * mp.rows * G + rel + i;
                        for (int r = 0; r < mp.rows; ++r) {
                            *dstp = out_conv<OUTMODE>((uint32_t)*srcp, norm_sum);
                            srcp += G;
                            dstp += o.row_stride;
                        }
                    } else {
                        const uint32_t *srcp = bins + slot * mp.rows * G + rel + i;
                        for (int r = 0; r < mp.rows; ++r) {
                            *dstp = out_conv<OUTMODE>(*srcp, norm_sum);
                            srcp += G;
                            dstp += o.row_stride;
                        }
                    }
                }
                continue;
            }
            if (B16) {   // (several rows: only summed slices come this way -- the laid-out ones took the branch above)
                for (int r = 0; r < mp.rows; ++r) {
                    const uint16_t *srcb = (const uint16_t *)bins + (slot * mp.rows + r) * G + rel;
                    typename OutT_<OUTMODE>::type *dst = out + o.out_off + (int64_t)r * o.row_stride;
                    if (o.step != 0) {
                        for (int i = i0 + lane0; i < i1; i += lanes) dst[(int64_t)o.step * i] = out_conv<OUTMODE>((uint32_t)srcb[i], norm_sum);
                    } else {
                        unsigned long long part = 0;
                        for (int i = i0 + lane0; i < i1; i += lanes) part += srcb[i];
                        out_add<OUTMODE>(dst, part);
                    }
                }
                continue;
            }
            for (int r = 0; r < mp.rows; ++r) {
                const uint32_t *srcb = bins + (slot * mp.rows + r) * G + rel;
                typename OutT_<OUTMODE>::type *dst = out + o.out_off + (int64_t)r * o.row_stride;
                if (o.step != 0) {
                    int i = i0 + (int)threadIdx.x;
                    for (; i + 3 * WG < i1; i += 4 * WG) { // four LDS reads in flight per lane
                        const uint32_t v0 = srcb[i], v1 = srcb[i + WG], v2 = srcb[i + 2 * WG], v3 = srcb[i + 3 * WG];
                        dst[(int64_t)o.step * i] = out_conv<OUTMODE>(v0, norm_sum);
                        dst[(int64_t)o.step * (i + WG)] = out_conv<OUTMODE>(v1, norm_sum);
                        dst[(int64_t)o.step * (i + 2 * WG)] = out_conv<OUTMODE>(v2, norm_sum);
                        dst[(int64_t)o.step * (i + 3 * WG)] = out_conv<OUTMODE>(v3, norm_sum);
                    }
                    for (; i < i1; i += WG) dst[(int64_t)o.step * i] = out_conv<OUTMODE>(srcb[i], norm_sum);
                } else {
                    unsigned long long part = 0;
                    for (int i = i0 + (int)threadIdx.x; i < i1; i += WG) part += srcb[i];
                    out_add<OUTMODE>(dst, part);
                }
            }
        }
    } else {
        // ---- split tile (pile-up or several files): merge into the zeroed compact histogram;
        // k_gather_split lays it out afterwards
        for (uint32_t pi = w.piece_begin; pi < w.piece_end; ++pi) {
            const Piece pc_ = pieces[pi];
            const int rel = pc_.start - w.win_start;
            for (int r = 0; r < mp.rows; ++r) {
                // (not c.base[pc_.mode]: a run-time index into that array would put the whole struct into
                // scratch memory -- 44 bytes per lane written to HBM by every work item)
                const int mode_base = __popc(w.mode_mask & ((1u << pc_.mode) - 1u)) * mp.rows * G;
                uint32_t *dst = hist + (size_t)r * hist_row_stride + pc_.hist_off;
                for (int i = threadIdx.x; i < pc_.len; i += WG) {
                    const uint32_t v = B16 ? (uint32_t)((const uint16_t *)bins)[mode_base + r * G + rel + i] : bins[mode_base + r * G + rel + i];
                    if (v) atomicAdd(&dst[i], v);
                }
            }
        }
    }
    if (SINGLE && done) {   // pc_query_segment: the counts are in the host's buffer -- say so, behind them
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(done, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---------------------------------------------------------------- k_gather_split
// Split tiles only: lay out their segment slices from the merged histogram, then clear the
// tile's histogram region so the next call starts from zeros again.  Last kernel of a call.  The
// work lists, their counters and the per-tile item counts belong to the plan and are left as they
// are: the next count of the plan serves the same lists without running k_tile_ranges again.
// `per_wg` tiles are looked at by one workgroup: 1 when every window is merged (several BAM
// files), 256 otherwise -- merged windows are then the exception (pile-ups), and a sparse
// annotation has hundreds of thousands of windows that would each cost an empty workgroup.
template <int OUTMODE>
__device__ __forceinline__ void gather_tile(const Tile &tl, const Piece *__restrict__ pieces,
                                            const OutPiece *__restrict__ opieces, int rows, uint32_t *hist,
                                            int64_t hist_row_stride, typename OutT_<OUTMODE>::type *out,
                                            double norm_sum) {
    for (uint32_t oi = tl.op_begin; oi < tl.op_end; ++oi) {
        const OutPiece o = opieces[oi];
        for (int r = 0; r < rows; ++r) {
            const uint32_t *src = hist + (size_t)r * hist_row_stride + o.hist_off;
            typename OutT_<OUTMODE>::type *dst = out + o.out_off + (int64_t)r * o.row_stride;
            if (o.step != 0) {
                for (int i = threadIdx.x; i < o.len; i += kWG) dst[(int64_t)o.step * i] = out_conv<OUTMODE>(src[i], norm_sum);
            } else {
                unsigned long long part = 0;
                for (int i = threadIdx.x; i < o.len; i += kWG) part += src[i];
                out_add<OUTMODE>(dst, part);
            }
        }
    }
    __syncthreads();
    for (uint32_t pi = tl.piece_begin; pi < tl.piece_end; ++pi) {
        const Piece pc_ = pieces[pi];
        for (int r = 0; r < rows; ++r) {
            uint32_t *dst = hist + (size_t)r * hist_row_stride + pc_.hist_off;
            for (int i = threadIdx.x; i < pc_.len; i += kWG) dst[i] = 0u;
        }
    }
}

// The compact histogram of a large plan is never cleared as a whole (C5: 4 GB, 0.6 ms of its first count): behind
// k_tile_ranges -- whenever the lists of a plan were (re)built -- this kernel zeroes the slices of the windows the lists
// merge (tile_items != 0); k_gather_split leaves them zero again after every count, and no other window's slice is ever
// read.  Sixty-four windows per workgroup: merged windows are the exception.
constexpr int kClearPerWG = 64;
__global__ __launch_bounds__(kWG) void k_clear_split(const Tile *__restrict__ tiles, int ntiles, const Piece *__restrict__ pieces,
                                                     const uint32_t *__restrict__ tile_items, const uint32_t *__restrict__ counters, int rows,
                                                     uint32_t *hist, int64_t hist_row_stride) {
    if (counters[4] == 0u) return;   // no window is merged
    __shared__ uint32_t s_list[kClearPerWG];
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0u;
    __syncthreads();
    const int t = (int)blockIdx.x * kClearPerWG + (int)threadIdx.x;
    if ((int)threadIdx.x < kClearPerWG && t < ntiles && tile_items[t] != 0u) s_list[atomicAdd(&s_n, 1u)] = (uint32_t)t;
    __syncthreads();
    const uint32_t n = s_n;
    for (uint32_t k = 0; k < n; ++k) {
        const Tile tl = tiles[s_list[k]];
        for (uint32_t pi = tl.piece_begin; pi < tl.piece_end; ++pi) {
            const Piece pc_ = pieces[pi];
            for (int r = 0; r < rows; ++r) {
                uint32_t *dst = hist + (size_t)r * hist_row_stride + pc_.hist_off;
                for (int i = threadIdx.x; i < pc_.len; i += kWG) dst[i] = 0u;
            }
        }
    }
}

template <int OUTMODE>
__global__ __launch_bounds__(kWG) void k_gather_split(const Tile *__restrict__ tiles, int ntiles, int per_wg,
                                                      const Piece *__restrict__ pieces,
                                                      const OutPiece *__restrict__ opieces,
                                                      const uint32_t *__restrict__ tile_items, const uint32_t *__restrict__ counters, int rows,
                                                      uint32_t *hist, int64_t hist_row_stride,
                                                      typename OutT_<OUTMODE>::type *out, double norm_sum,
                                                      uint32_t launched_heavy, uint32_t launched_light, uint32_t launched_small,
                                                      uint32_t work_cap, uint32_t *grid_error) {
    __shared__ uint32_t s_list[kWG];
    __shared__ uint32_t s_n;
    if (blockIdx.x == 0 && threadIdx.x < 4) {
        // guard of the exact grids: the histogram kernels were launched with the work counts a previous count of
        // this plan left (0xffffffff: the whole capacity was launched); if this count queued more, items went unserved
        const uint32_t launched = threadIdx.x == 0 ? launched_heavy : (threadIdx.x == 1 ? launched_light : (threadIdx.x == 2 ? launched_small : 0xffffffffu));
        if (launched != 0xffffffffu && counters[threadIdx.x] > launched) atomicOr(grid_error, 1u);
        // guard of the list capacity: heavy items fill the list from the front, light ones from the back; more of them
        // than slots means the two ends overwrote each other (k_tile_ranges only keeps its writes inside the list)
        if (threadIdx.x == 0 && (uint64_t)counters[0] + (uint64_t)counters[1] > (uint64_t)work_cap) atomicOr(grid_error, 2u);
    }
    if (per_wg == 1) {
        if (tile_items[blockIdx.x] == 0u) return; // only windows that were merged through the histogram
        gather_tile<OUTMODE>(tiles[blockIdx.x], pieces, opieces, rows, hist, hist_row_stride, out, norm_sum);
        return;
    }
    const int t = (int)blockIdx.x * kWG + (int)threadIdx.x;
    if (threadIdx.x == 0) s_n = 0u;
    __syncthreads();
    if (t < ntiles && tile_items[t] != 0u) s_list[atomicAdd(&s_n, 1u)] = (uint32_t)t;
    __syncthreads();
    const uint32_t n = s_n;
    for (uint32_t k = 0; k < n; ++k) { // rare; the whole workgroup lays each of them out in turn
        gather_tile<OUTMODE>(tiles[s_list[k]], pieces, opieces, rows, hist, hist_row_stride, out, norm_sum);
        __syncthreads();
    }
}

// ---------------------------------------------------------------- k_center
// CenterMapFactory: count[p] is the left-to-right float64 sum, in read order, of
// 1/(L-2*nibble) over the reads whose trimmed positions contain p.  The order is
// part of the contract (the reference's own test demands exact equality), so
// there are no atomics and no re-association: one lane owns one output position and
// replays, in record order, every read that can cover it.
//
// Round 4: FOUR ordered replays per wave, one per 16-lane row.  A wave still owns a chunk of <= 64 positions, but every
// row of 16 lanes owns 16 of them and walks ITS OWN stretch of the center stream (below): the entries of the records
// that start in [row start - W + 1, row end).  A row of 16 positions has to look at (16 + W - 1) positions' worth of
// reads, a wave of 64 at (64 + W - 1) -- with 30-nt reads that is 50 against 98 -- and the four rows advance together:
// one step applies entry j of every row to that row's lanes.  Every lane prepares ONE entry per batch of 16 steps:
// its 16-bit coverage mask of the row (bit i: position i of the row is counted) and half its value, 0.5 / (L - 2 nibble).
// The entry is then broadcast INSIDE its row by DPP (row_newbcast:j -- lane j of each row to all 16 lanes of the
// row), fused into the instructions that use it, so a step is three vector instructions and nothing else:
//     v_and_b32_dpp   x, cm, lbit          x      = cm[j] & (1 << lane)        covered: 1 << lane, else 0
//     v_lshlrev_b32   one.hi, 30 - lane, x   one = covered ? 2.0 : 0.0           (0x40000000 is the high word of 2.0)
//     v_fmac_f64_dpp  acc, valh, one       acc    = fma(valh[j], one, acc)
// fma(val / 2, 2.0, acc) rounds val + acc once -- the reference's `count[c] += val` bit for bit (halving and doubling
// are exact) -- and fma(val / 2, 0.0, acc) is acc: the conditional add without touching exec.  No scalar
// instruction, no LDS, no v_readlane in the loop.  Why this form (scripts/ubench/valu_cost_probe.hip, center_dpp_probe.hip,
// cycles per instruction and SIMD at eight waves): only plain 32-bit VOP1/VOP2 instructions issue at 2 cycles; DPP,
// SDWA, VOP3 encodings, an SGPR or carry operand cost 4 (a DPP on an f64 instruction is free), the CU's ONE scalar
// unit serves each SIMD every 4 cycles -- round 3's replay (3 SALU + 3.25 VALU + the add per entry) ran at 19.5 cycles
// per entry, a compare-and-select step (sub, borrow, cndmask, fmac) at 21, this one at 12.  (A DPP *rev* opcode
// broadcasts its SECOND source on gfx950 -- v_subrev_u32_dpp d, a, b = dpp(b) - a, dpp_semantics.hip -- the AND is
// commutative.)

// wave-uniform value of lane `j` of a per-lane register (j uniform)
__device__ __forceinline__ uint32_t lane_u32(uint32_t v, int j) { return (uint32_t)__builtin_amdgcn_readlane((int)v, j); }
__device__ __forceinline__ double lane_f64(double v, int j) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = lane_u32((uint32_t)b, j), hi = lane_u32((uint32_t)(b >> 32), j);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Workgroup of the center kernel: ONE wave.  The waves of a launch are independent (one chunk each) and very
// unequal; in a four-wave workgroup the wave slots of the finished ones stay taken until the last one is done.
#ifndef PC_CENTER_WG
#define PC_CENTER_WG 64
#endif
constexpr int kCenterWG = PC_CENTER_WG;

// Center stream of a staged file and a strand selection (0: forward reads, 1: reverse reads, 2: all reads), built
// on the GPU at the first center-rule count of the file and again when the host-side filters change:
//   cs_ent[k] = {x, y}: one entry per aligned run of a read, ALREADY TRIMMED by the rule's nibble: x = first counted
//               position of the run, y = counted positions m (0: the nibble leaves nothing of this run) | aligned length
//               L of the read << 16 | flags << 24.  (CenterMapFactory counts read indices [nibble, L - nibble),
//               map_factories.pyx:250-254; a run whose first base is read index `cum` keeps [max(cum, nibble),
//               min(cum + len, L - nibble)).)  A read contributes its runs consecutively, reads in record order; reads the
//               selection drops, host-excluded reads and reads without aligned bases have no entry.  flags bit 0
//               (kCsIndirect): the read does not fit the 8-bit fields (L > 255) -- one entry, x = its record index.
//               The entries depend on the nibble: a change of the rule's parameter re-runs k_cs_scatter (not the scan).
//   cs_soff[i] = entries before record i (cs_soff[n] = all): turns a record range into an entry range.
constexpr uint32_t kCsIndirect = 1u;

__device__ __forceinline__ uint32_t cs_entries_of(uint32_t meta, int sel) {
    const uint32_t fl = rec_flags(meta);
    if ((fl & kFlagExcluded) || rec_len(meta) == 0) return 0u;
    if (sel < 2 && (int)(fl & kFlagReverse) != sel) return 0u;
    return rec_len(meta) > 255 ? 1u : (uint32_t)(rec_nblk(meta) >= 2 ? rec_nblk(meta) : 1);
}

__global__ __launch_bounds__(kWG) void k_cs_count(const uint2 *__restrict__ rec, int64_t n, int sel, uint32_t *cnt) {
    const int64_t i = (int64_t)blockIdx.x * kWG + threadIdx.x;
    if (i > n) return;
    cnt[i] = i < n ? cs_entries_of(rec[i].y, sel) : 0u;   // entry n: the exclusive sum then ends with the total
}

__global__ __launch_bounds__(kWG) void k_cs_scatter(const uint2 *__restrict__ rec, const uint32_t *__restrict__ blk_off,
                                                    const int2 *__restrict__ blk, int64_t n, int sel, int nib,
                                                    const uint32_t *__restrict__ soff, uint2 *ent) {
    const int64_t i = (int64_t)blockIdx.x * kWG + threadIdx.x;
    if (i >= n) {   // padding behind the last entry: entries that cover nothing (whole batches can always be loaded)
        if (i < n + 64) ent[(int64_t)soff[n] + (i - n)] = make_uint2(0x7fffffffu, 0u);
        return;
    }
    const uint2 r = rec[i];
    const uint32_t k = cs_entries_of(r.y, sel);
    if (k == 0u) return;
    uint2 *dst = ent + soff[i];
    const uint32_t L = (uint32_t)rec_len(r.y);
    if (L > 255u) { dst[0] = make_uint2((uint32_t)i, kCsIndirect << 24); return; }
    // run [start, start + len) holding read indices [cum, cum + len): what the nibble leaves of it
    auto entry = [&](int32_t start, int len, int cum) {
        const int lo_i = cum > nib ? cum : nib, hi_i = cum + len < (int)L - nib ? cum + len : (int)L - nib;
        const int m = hi_i > lo_i ? hi_i - lo_i : 0;
        return make_uint2((uint32_t)(start + (lo_i - cum)), (uint32_t)m | (L << 16));
    };
    if (rec_nblk(r.y) < 2) { dst[0] = entry((int32_t)r.x, (int)L, 0); return; }
    const int2 *b = blk + blk_off[i];
    int cum = 0;
    for (uint32_t q = 0; q < k; ++q) {
        const int2 run = b[q];
        dst[q] = entry(run.x, run.y, cum);
        cum += run.y;
    }
}

// half the value of a read by aligned length, for the lengths a stream entry can carry: 0.5 / (L - 2 nibble), or 0.0
// where the read is not counted (size filter, nothing left by the nibble: adding +0.0 never changes a sum)
__global__ void k_center_vals(MapParams mp, const double *__restrict__ invh, double *cvalh) {
    const int L = (int)threadIdx.x, m = L - 2 * mp.param;
    if (L < 256) cvalh[L] = (m > 0 && size_ok(mp, L)) ? invh[m] : 0.0;
}

// Dispatch list of the center kernel.  A chunk's replay is sequential in the reads that overlap
// it (the order is the contract), so its time is proportional to that count, and expression is
// heavy-tailed: the kernel would wait for the chunks over the deepest pile-up.  Two measures:
//   * chunks with many candidates are CUT into 4 or 8 sub-chunks of 16 / 8 positions, one wave
//     each, whose rows own 4 / 2 positions: a row then replays the reads over (4 + W) or (2 + W) instead of
//     (16 + W) positions -- the critical path shrinks for idle lanes in a few waves;
//   * those entries are queued first (longest-first in two classes, as for the histogram work
//     list): list[0 .. nheavy) heavy, list[cap-1 .. cap-nlight] light, cap = kCenterCap * nchunks.
// Thresholds are relative to the mean candidate count (pass 1 sums it), so by Markov's
// inequality fewer than nchunks/8 chunks are cut and the heavy entries fit in nchunks slots.
// Entry = chunk index | code << 27: 0 whole chunk, 1..4 quarter, 5..12 eighth.
constexpr int kSubShift = 27;
constexpr uint32_t kCenterCap = 3u;   // dispatch-list slots per chunk: front entries < (8/k1 + 1/8) x chunks by Markov, the rest from the back

__device__ __forceinline__ int center_sel(int mode) { return mode < 2 ? mode : 2; }
// (indexed in memory: a run-time index into a register copy of the view would put that copy into scratch)
__device__ __forceinline__ const u32x2 PC_GLOBAL *cs_stream(const FileView *f, int sel) { return (const u32x2 PC_GLOBAL *)f->cs_ent[sel]; }
__device__ __forceinline__ const uint32_t PC_GLOBAL *cs_offsets(const FileView *f, int sel) { return (const uint32_t PC_GLOBAL *)f->cs_soff[sel]; }

// pass 1 (one THREAD per chunk): per file the EXACT record range of the chunk's near window -- first record that
// starts at or after start - W + 1, first record that starts at or after the chunk's end (index bucket, then a
// bisection inside it) -- the entry ranges of its four rows of 16 positions (bisections inside that record range),
// and the candidate range of the long-span list; all the dependent index lookups happen
// here, once, instead of at the head of every wave of k_center.  The candidate count (stream entries of the near
// window) and its sum (counters[2..3] as one 64-bit value) feed the dispatch order.
__device__ __forceinline__ int64_t bucket_lower_bound(const GFile &fv, int64_t q0, int64_t nb, int64_t key) {
    int64_t b = key <= 0 ? 0 : (key >> kLinShift);
    if (b > nb) b = nb;
    const int64_t b1 = b + 1 > nb ? nb : b + 1;
    return lower_bound_pos(fv.rec, fv.lin_tab[q0 + b], fv.lin_tab[q0 + b1], key);
}

constexpr int kCenterRows = 4;        // 16-lane rows of a wave: independent replays

__global__ __launch_bounds__(kRangesWG) void k_center_weigh(const CenterChunk *__restrict__ chunks, int64_t nchunks,
                                                            const FileView *__restrict__ files, int nfiles, int W,
                                                            uint32_t *cand_out, u32x4 *ranges, u32x2 *rec_ranges, uint32_t *row_ranges,
                                                            unsigned long long *total) {
    const int64_t c = (int64_t)blockIdx.x * kRangesWG + threadIdx.x;
    unsigned long long cand = 0;
    if (c < nchunks) {
        const CenterChunk ck = chunks[c];
        for (int f = 0; f < nfiles; ++f) {
            const GFile fv = gfile(files[f]);
            const int64_t q0 = fv.lin_off[ck.tid], nb = fv.lin_off[ck.tid + 1] - q0 - 1;
            const int64_t cend = (int64_t)ck.start + ck.len;
            u32x4 rg;
            u32x2 rr;   // the near window as a record range (what a sub-chunk narrows) and as the entry range of the chunk's stream
            rr.x = (uint32_t)bucket_lower_bound(fv, q0, nb, (int64_t)ck.start - W + 1);
            rr.y = (uint32_t)bucket_lower_bound(fv, q0, nb, cend);
            if (rr.y < rr.x) rr.y = rr.x;
            const uint32_t PC_GLOBAL *soff = cs_offsets(files + f, center_sel(ck.mode));
            rg.x = soff[rr.x];
            rg.y = soff[rr.y];
            rg.z = rg.w = 0u;
            if (fv.nlong) { // they start before the near window of some position of the chunk, and the
                            // running maximum of the ends has passed the chunk start
                rg.z = (uint32_t)lin_floor(fv.plin_tab, q0, nb, ck.start);
                rg.w = (uint32_t)lin_floor(fv.llin_tab, q0, nb, cend - W + (1 << kLinShift) - 1);
                if (rg.z > rg.w) rg.z = rg.w;
            }
            ranges[c * nfiles + f] = rg;
            rec_ranges[c * nfiles + f] = rr;
            // rows: [first entry of the records that start at or after row start - W + 1, ... at or after row end)
            uint32_t *rt = row_ranges + (size_t)(c * nfiles + f) * (2 * kCenterRows);
            int64_t from = rr.x;
            for (int r = 0; r < kCenterRows; ++r) {
                const int64_t rs = (int64_t)ck.start + 16 * r;
                if (rs >= cend) { rt[r] = rg.y; continue; }
                from = r == 0 ? (int64_t)rr.x : lower_bound_pos(fv.rec, from, rr.y, rs - W + 1);
                rt[r] = soff[from];
            }
            from = rr.x;
            for (int r = 0; r < kCenterRows; ++r) {
                const int64_t rs = (int64_t)ck.start + 16 * r, re = rs + 16 < cend ? rs + 16 : cend;
                if (rs >= cend) { rt[kCenterRows + r] = rg.y; continue; }
                from = re >= cend ? (int64_t)rr.y : lower_bound_pos(fv.rec, from, rr.y, re);
                rt[kCenterRows + r] = soff[from];
            }
            cand += rg.y - rg.x;
        }
        cand_out[c] = (uint32_t)(cand > 0xffffffffull ? 0xffffffffull : cand);

    }
    for (int o = 32; o > 0; o >>= 1) cand += __shfl_down(cand, o, 64);
    if ((threadIdx.x & 63) == 0 && cand) atomicAdd(total, cand);
}

// pass 2: cut and queue
__global__ __launch_bounds__(kRangesWG) void k_center_order(const uint32_t *__restrict__ cand_in, int64_t nchunks,
                                                            const unsigned long long *__restrict__ total,
                                                            int64_t floor_thr, int64_t floor_whole, int k1, int k2,
                                                            uint32_t *order, uint32_t *counters) {
    __shared__ uint32_t s_wave[kRangesWG / 64];
    __shared__ uint32_t s_base[2];
    const int64_t c = (int64_t)blockIdx.x * kRangesWG + threadIdx.x;
    const bool live = c < nchunks;
    const int64_t mean = (int64_t)(*total / (unsigned long long)(nchunks > 0 ? nchunks : 1));
    const int64_t t1 = k1 * mean > floor_thr ? k1 * mean : floor_thr, t2 = k2 * t1; // k1 >= 8 (capacity, see above)
    const int64_t th_whole = 8 * mean > floor_whole ? 8 * mean : floor_whole;       // starts early, but stays whole
    const int64_t cand = live ? (int64_t)cand_in[c] : 0;
    // entries a chunk puts at the FRONT of the list: 8 / 4 sub-chunks, 1 = the whole chunk (heavy but
    // below the cut threshold: cutting multiplies the work -- eight sub-chunks scan 8 x 4 x (2 + W)
    // positions' worth of reads instead of 4 x (16 + W) -- so only the deepest pile-ups, whose sequential
    // replay would otherwise outlast the rest of the launch, are cut); 0 = light, queued from the back
    const uint32_t nsub = !live ? 0u : (cand > t2 ? 8u : (cand > t1 ? 4u : (cand > th_whole ? 1u : 0u)));
    uint32_t th, tl;
    const uint32_t oh = block_scan_excl(nsub, s_wave, th);
    const uint32_t ol = block_scan_excl((live && nsub == 0u) ? 1u : 0u, s_wave, tl);
    if (threadIdx.x == 0) {
        s_base[0] = th ? atomicAdd(&counters[0], th) : 0u;
        s_base[1] = tl ? atomicAdd(&counters[1], tl) : 0u;
    }
    __syncthreads();
    if (!live) return;
    const uint32_t cap = kCenterCap * (uint32_t)nchunks;
    if (nsub == 0u) {
        order[cap - 1u - (s_base[1] + ol)] = (uint32_t)c;
    } else if (nsub == 1u) {
        order[s_base[0] + oh] = (uint32_t)c;
    } else {
        const uint32_t first_code = nsub == 4u ? 1u : 5u;
        for (uint32_t k = 0; k < nsub; ++k) order[s_base[0] + oh + k] = (uint32_t)c | ((first_code + k) << kSubShift);
    }
}

// One read replayed from its record header (wave-uniform arguments): reads that come from the long-span list with more
// than two runs, and reads the 8-bit fields of a stream entry cannot describe.  CenterMapFactory.__call__,
// map_factories.pyx:242-256.
__device__ __forceinline__ void center_read(const GFile &fv, const MapParams &mp, const double PC_GLOBAL *inv, int32_t pos,
                                            int L, int nbk, uint32_t boff, int32_t p, double &acc) {
    const int nib = mp.param;
    const int m = L - 2 * nib;                               // map_length, :245
    if (m <= 0 || !size_ok(mp, L)) return;
    const double val = m < 65536 ? inv[m] : 1.0 / (double)m; // 1.0 / map_length, :250 (the table holds the common lengths)
    bool hit;
    if (nbk < 2) {
        hit = (uint32_t)(p - (pos + nib)) < (uint32_t)m;
    } else {                                                 // positions with read index in [nib, L - nib)
        hit = false;
        int cum = 0;
        for (int q = 0; q < nbk; ++q) {
            const i32x2 run = fv.blk[boff + q];
            const int idx = cum + (p - run.x);
            hit |= (p >= run.x) & (p < run.x + run.y) & (idx >= nib) & (idx < L - nib);
            cum += run.y;
        }
    }
    acc += hit ? val : 0.0;                                  // :254, one IEEE add per covering read, in order
}

// The replay steps.  Operands: acc (the lane's sum), cm / valh (the entry this lane prepared: entry `lane & 15` of its
// row's batch), lbit = 1 << (lane & 15), sh = 30 - (lane & 15).  Temporaries are fixed registers (the high half of `one`
// has to be named on its own): v8, v9 = x of two steps in flight, v[10:11], v[12:13] = their `one` (low halves zero).
// The DPP operands (cm, valh) are never written inside a block; the two v_mov and the s_nop at its head keep the
// producer of an operand two instructions away from its first DPP read (the hazard the compiler cannot see through
// inline asm).
#define PC_CS_AND(J, T) "v_and_b32_dpp " T ", %[cm], %[lbit] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
#define PC_CS_SHL(OH, T) "v_lshlrev_b32 " OH ", %[sh], " T "\n\t"
#define PC_CS_FMA(J, PAIR) "v_fmac_f64_dpp %[acc], %[val], " PAIR " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
#define PC_CS_2(J0, J1)                                                                                                \
    PC_CS_AND(J0, "v8") PC_CS_AND(J1, "v9") PC_CS_SHL("v11", "v8") PC_CS_SHL("v13", "v9")                               \
    PC_CS_FMA(J0, "v[10:11]") PC_CS_FMA(J1, "v[12:13]")
#define PC_CS_HEAD "v_mov_b32 v10, 0\n\tv_mov_b32 v12, 0\n\ts_nop 1\n\t"
#define PC_CS_Q0 PC_CS_2(0, 1) PC_CS_2(2, 3)
#define PC_CS_Q1 PC_CS_2(4, 5) PC_CS_2(6, 7)
#define PC_CS_Q2 PC_CS_2(8, 9) PC_CS_2(10, 11)
#define PC_CS_Q3 PC_CS_2(12, 13) PC_CS_2(14, 15)
#define PC_CENTER_STEPS(CODE)                                                                                          \
    asm volatile(PC_CS_HEAD CODE                                                                                       \
                 : [acc] "+v"(acc)                                                                                     \
                 : [cm] "v"(cm_), [val] "v"(valh_), [lbit] "v"(lane_bit), [sh] "v"(lane_sh)                            \
                 : "v8", "v9", "v10", "v11", "v12", "v13")

// Experiment hook (scripts/exp_center_sections.py; never set in the product build): a build with -DPC_CENTER_SKIP=<mask>
// leaves out 1 the replay steps (entries are still loaded and prepared), 2 the near-window stream loop altogether,
// 4 the epilogue (output pieces, stores), 8 the per-wave LDS copy of the by-length table -- each section's share of
// k_center is the difference in time (the results are then wrong).
#ifndef PC_CENTER_SKIP
#define PC_CENTER_SKIP 0
#endif

// ---------------------------------------------------------------- k_center_slots / k_center2 (round 5)
// Where round 4's k_center spent its time (scripts/exp_center_sections.py, sections compiled out): of 1.20 ms on C3 the
// replay steps themselves are 0.14 -- the rest is LATENCY: a wave's four dependent loads before its first entry
// (counters -> order -> {chunk, ranges, row ranges} -> entries: 0.36 ms for 372 k waves with nothing else to do), the
// entry loop at two batches in flight (0.42 ms without a single step), and the output pieces fetched after the last step
// (0.2 - 0.3 ms).  A wave spent 14 of its 26 us outside the replay loop, so on average fewer than four of a SIMD's eight
// waves had steps to issue.  Hence (round 5 for plans over ONE alignment file; since round 6 for every plan -- a plan
// over several files has one descriptor per entry and FILE, consecutive, replayed into the same sums in file order):
//   * k_center_slots (cached with the dispatch list: once per plan, alignments and halo) resolves every dispatch entry
//     into a 128-byte DESCRIPTOR: the (sub-)chunk's positions, its rows' entry ranges (those of a sub-chunk's narrower
//     rows by the bisections the wave used to make), the long-span candidates, and -- when exactly one output piece
//     takes the chunk's sums, the common case -- that piece inline;
//   * k_center2: a wave loads a descriptor with ONE coalesced 128-byte request (lane l holds dword l; fields come out
//     by v_readlane, the rows' ranges by a lane permute) and requests its entries right behind it: two dependent trips
//     instead of four, none in the epilogue;
//   * a wave serves PC_CENTER_PER_WAVE consecutive light entries (heavy ones -- long replays -- keep a wave each) and has
//     the NEXT descriptor in flight while it replays: the launch has a quarter of the waves, and three of four chunks
//     start with their descriptor already in registers;
//   * PC_CENTER2_RING batches of entries in flight per wave, and no load is issued for a batch behind the longest row.
struct CenterSlot {
    int32_t start;             // first position of the (sub-)chunk
    uint32_t shape;            // positions (0: nothing to do) | row width << 8 | strand mode << 16 | 1 << 24: `out_off` .. `ostep` hold THE output piece
    uint32_t ent0;             // first stream entry of the chunk's near window: base of the 32-bit entry offsets below
    uint32_t to_end;           // entries from there to the end of the stream (64 entries that cover nothing follow it)
    uint32_t lo[kCenterRows], hi[kCenterRows];   // the rows' entry ranges, relative to ent0
    uint32_t long_z, long_w;   // candidate range of the long-span list
    uint32_t nmax;             // entries of the longest row
    uint32_t chunk;            // chunk index | code << kSubShift (diagnostics)
    uint32_t out_lo, out_hi;   // out_off of the inline output piece (two halves: the descriptor is read dword by dword)
    int32_t ostart, olen, ostep;
    uint32_t op_begin, op_end; // the window's output pieces: the general epilogue when no piece, or several, take the sums
    int32_t tid;
    uint32_t pad[8];
};
static_assert(sizeof(CenterSlot) == 128, "a descriptor is 32 dwords: one per lane of half a wave");
enum { kCsStart = 0, kCsShape = 1, kCsEnt0 = 2, kCsToEnd = 3, kCsLo = 4, kCsHi = 8, kCsLongZ = 12, kCsLongW = 13, kCsNmax = 14,
       kCsChunk = 15, kCsOutLo = 16, kCsOutHi = 17, kCsOStart = 18, kCsOLen = 19, kCsOStep = 20, kCsOpBegin = 21, kCsOpEnd = 22, kCsTid = 23 };

#ifndef PC_CENTER_PER_WAVE
#define PC_CENTER_PER_WAVE 1   // light entries per wave (4 / 8 measured on C3: the bulk of the launch ends no earlier, and a wave over 4 consecutive dense chunks becomes its tail: 1.85 / 2.71 ms against 1.18)
#endif
#ifndef PC_CENTER_HEAVY_PRIO
#define PC_CENTER_HEAVY_PRIO 3  // s_setprio of the waves that serve heavy entries (0: none)
#endif
#ifndef PC_CENTER2_RING
#define PC_CENTER2_RING 4
#endif
#ifndef PC_CENTER_HEAVY_RING
#define PC_CENTER_HEAVY_RING 12   // batches in flight for a heavy entry
#endif

// steps the replay executes for a row range of `n` entries (batches of 16, the last one rounded up to whole quarters)
__device__ __forceinline__ uint32_t center_steps_of(uint32_t n) { return (n & ~15u) + (((n & 15u) + 3u) & ~3u); }

// One THREAD per dispatch entry (heavy entries first, then the light ones in list order): its descriptor.
// fill[0] += entries of all rows, fill[1] += 4 x replay steps (the lock-step rows' capacity): their ratio is the row
// fill the bench line reports.
__global__ __launch_bounds__(kRangesWG) void k_center_slots(const CenterChunk *__restrict__ chunks, int64_t nchunks, const FileView *__restrict__ files, int nfiles,
                                                            int W, const uint32_t *__restrict__ order, const uint32_t *__restrict__ counters,
                                                            const u32x4 *__restrict__ ranges, const u32x2 *__restrict__ rec_ranges,
                                                            const uint32_t *__restrict__ row_ranges, const OutPiece *__restrict__ opieces,
                                                            CenterSlot *slots, unsigned long long *fill) {
    const uint32_t n_heavy = counters[0], n_light = counters[1];
    const uint32_t e = blockIdx.x * (uint32_t)kRangesWG + threadIdx.x;
    unsigned long long f_ent = 0, f_cap = 0;
    if (e < n_heavy + n_light) {
        const uint32_t cap = kCenterCap * (uint32_t)nchunks;
        const uint32_t entry = e < n_heavy ? order[e] : order[cap - 1u - (e - n_heavy)];
        const uint32_t cidx = entry & ((1u << kSubShift) - 1u), code = entry >> kSubShift;
        const CenterChunk ck = chunks[cidx];
        // (several files: the descriptors of one entry are consecutive, file-major -- the order the reference's fetch
        // chains the files in, genome_array.py:800-809 -- and every one carries the output piece)
        for (int f = 0; f < nfiles; ++f) {
        CenterSlot sl;
        for (int k = 0; k < 8; ++k) sl.pad[k] = 0u;
        const int sub_off = code == 0u ? 0 : (code <= 4u ? 16 * (int)(code - 1u) : 8 * (int)(code - 5u));
        const int roww = code == 0u ? 16 : (code <= 4u ? 4 : 2);
        const int32_t s0 = ck.start + sub_off;
        const int32_t cend = ck.start + (ck.len < sub_off + kCenterRows * roww ? ck.len : sub_off + kCenterRows * roww);
        const int npos = cend > s0 ? cend - s0 : 0;
        const int sel = center_sel(ck.mode);
        const u32x4 rg = ((const u32x4 PC_GLOBAL *)ranges)[(size_t)cidx * nfiles + f];
        sl.start = s0;
        sl.ent0 = rg.x;
        sl.to_end = files[f].cs_total[sel] - rg.x;
        sl.long_z = rg.z; sl.long_w = rg.w;
        sl.chunk = entry;
        sl.tid = ck.tid;
        sl.op_begin = ck.op_begin; sl.op_end = ck.op_end;
        sl.out_lo = sl.out_hi = 0u; sl.ostart = 0; sl.olen = 0; sl.ostep = 0;
        uint32_t nmax = 0;
        if (npos > 0) {
            const GFile fv = gfile(files[f]);
            const uint32_t PC_GLOBAL *soff = cs_offsets(files + f, sel);
            const u32x2 rr = ((const u32x2 PC_GLOBAL *)rec_ranges)[(size_t)cidx * nfiles + f];
            for (int r = 0; r < kCenterRows; ++r) {
                uint32_t lo, hi;
                if (code == 0u) {
                    const uint32_t PC_GLOBAL *rt = (const uint32_t PC_GLOBAL *)row_ranges + ((size_t)cidx * nfiles + f) * (2 * kCenterRows);
                    lo = rt[r]; hi = rt[kCenterRows + r];
                } else {   // the narrower rows of a sub-chunk: records that start in [row start - W + 1, row end)
                    const int32_t rs = s0 + r * roww, re = rs + roww < cend ? rs + roww : cend;
                    const bool row_live = rs < cend;
                    const int64_t key_lo = row_live ? (int64_t)rs - W + 1 : (int64_t)cend, key_hi = row_live ? (int64_t)re : (int64_t)cend;
                    const int64_t r0 = lower_bound_pos(fv.rec, rr.x, rr.y, key_lo);
                    const int64_t r1 = lower_bound_pos(fv.rec, r0, rr.y, key_hi);
                    lo = soff[r0]; hi = soff[r1];
                }
                if (hi < lo) hi = lo;
                sl.lo[r] = lo - rg.x; sl.hi[r] = hi - rg.x;
                nmax = hi - lo > nmax ? hi - lo : nmax;
                f_ent += hi - lo;
            }
            f_cap += (unsigned long long)kCenterRows * center_steps_of(nmax);
            // the output pieces that take this (sub-)chunk's sums: exactly one -- inline
            uint32_t hits = 0, which = 0;
            for (uint32_t oi = ck.op_begin; oi < ck.op_end; ++oi) {
                const OutPiece o = opieces[oi];
                if (o.mode == ck.mode && o.start < cend && o.start + o.len > s0) { ++hits; which = oi; }
            }
            if (hits == 1u) {
                const OutPiece o = opieces[which];
                sl.out_lo = (uint32_t)((unsigned long long)o.out_off); sl.out_hi = (uint32_t)((unsigned long long)o.out_off >> 32);
                sl.ostart = o.start; sl.olen = o.len; sl.ostep = o.step;
            }
            sl.shape = (uint32_t)npos | ((uint32_t)roww << 8) | ((uint32_t)ck.mode << 16) | (hits == 1u ? 1u << 24 : 0u);
            if (hits == 0u) sl.op_begin = sl.op_end = 0u;   // (nothing takes the sums: the positions are replayed for nobody -- cannot happen for a chunk cut from queried pieces, kept harmless)
        } else {
            for (int r = 0; r < kCenterRows; ++r) { sl.lo[r] = 0u; sl.hi[r] = 0u; }
            sl.shape = 0u;
        }
        sl.nmax = nmax;
        slots[(size_t)e * nfiles + f] = sl;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { f_ent += __shfl_down(f_ent, o, 64); f_cap += __shfl_down(f_cap, o, 64); }
    if ((threadIdx.x & 63) == 0 && f_cap) { atomicAdd(&fill[0], f_ent); atomicAdd(&fill[1], f_cap); }
}

struct Center2Ctx {
    const CenterSlot *slots;
    const uint2 *ent[3];       // the file's center streams by strand selection (kernel arguments: no load between descriptor and entries)
    uint32_t indirect;         // bit sel: that stream holds indirect entries (reads beyond the 8-bit fields)
    const FileView *files;
    int nfiles;                // files of the plan's engine: a dispatch entry has one descriptor per file (consecutive)
    FileView file0;            // the one file's view, by value: its pointers come out of the argument segment (scalar loads at the point of use) instead of per-lane loads from memory
    MapParams mp;
    int W;
    const double *inv, *invh, *cvalh;
    const uint32_t *counters;
    uint32_t known, n_heavy, n_light;   // known: the host has read the list's counts back (else the kernel reads `counters`)
    const OutPiece *opieces;
    double *out;
    double norm_sum;
    int norm_on;
    unsigned long long *dbg;
    uint32_t dbg_cap;
};

// One (sub-)chunk from its descriptor `d` (lane l < 32 holds dword l; the upper half of the wave holds a copy).
// `dsel`: 0 / 32 -- which half of the wave holds the descriptor (-1: both hold a copy).  RING: batches of entries in flight.
// MULTI: a plan over several alignment files -- the wave calls this once per file, in file order, with the same `acc`
// (the reads of file f + 1 follow those of file f in every position's sum: itertools.chain over the files' fetches,
// genome_array.py:800-809); the sums are written after the last file (`last`).  The file's view and streams then come
// from memory (cx.files[file]); a single file's travel in the kernel arguments.
template <bool DBG, bool GENERAL, int RING, bool MULTI = false>
__device__ __forceinline__ void center_slot(const Center2Ctx &cx, const uint32_t d_, const int dsel, const int lane, const double *s_valh, unsigned long long &n_slots,
                                            double &acc, const int file = 0, const bool last = true) {
    // (the descriptor in both halves of the wave, as the field reads below expect it; dsel < 0: it already is)
    const uint32_t d = dsel < 0 ? d_ : (uint32_t)__shfl((int)d_, dsel + (lane & 31), 64);
    const uint32_t shape = lane_u32(d, kCsShape);
    const int npos = (int)(shape & 0xffu);
    if (npos == 0) return;
    const int roww = (int)((shape >> 8) & 0xffu), mode = (int)((shape >> 16) & 0xffu);
    const double PC_GLOBAL *inv = (const double PC_GLOBAL *)cx.inv;      // 1.0 / m
    const double PC_GLOBAL *invh = (const double PC_GLOBAL *)cx.invh;    // 0.5 / m = (1.0 / m) / 2, exactly
    const MapParams &mp = cx.mp;
    const int W = cx.W;
    const int row = lane >> 4, li = lane & 15;
    const int nib = mp.param;
    const int32_t s0 = (int32_t)lane_u32(d, kCsStart), cend = s0 + npos;
    const int32_t rs = s0 + row * roww;                              // this lane's row: positions [rs, re)
    const bool row_live = rs < cend;
    const int32_t p = rs + li;
    const bool owns = li < roww && p < cend;
    const int sel = center_sel(mode);
    const int lane_bit = 1 << li, lane_sh = 30 - li;
    const GFile fv = MULTI ? gfile(cx.files[file]) : gfile(cx.file0);
    auto row_mask = [&](int a0, int m) {   // (m = 0: nothing)
        const int first = a0 - rs, b0 = first > 0 ? first : 0, b1 = first + m < 16 ? first + m : 16;   // row-relative [b0, b1)
        return b1 > b0 ? (int)((1u << b1) - (1u << b0)) : 0;
    };
    // one batch: entry `li` of every row in (cm_, valh_); `indirect`: a read the entry cannot describe (record `recidx`)
    auto replay = [&](int a0_, int mm_, int cm_, double valh_, bool indirect, uint32_t recidx, int nsteps, bool may_be_indirect) {
        if (DBG) n_slots += (unsigned long long)nsteps;
        if (may_be_indirect && __any(indirect)) {
            for (int j = 0; j < 16; ++j) {   // entry by entry, row by row (what matters is the order inside a row)
                for (int r = 0; r < kCenterRows; ++r) {
                    const int src = r * 16 + j;
                    if (lane_u32((uint32_t)indirect, src)) {
                        const int64_t i = (int64_t)lane_u32(recidx, src);
                        const u32x2 rr = fv.rec[i];
                        int Li, nbi;
                        rec_true(fv, i, rr.y, Li, nbi);
                        double t = acc;
                        center_read(fv, mp, inv, (int32_t)rr.x, Li, nbi, nbi >= 2 ? fv.blk_off[i] : 0u, p, t);
                        acc = row == r ? t : acc;
                    } else {
                        const int aj = (int)lane_u32((uint32_t)a0_, src), mj = (int)lane_u32((uint32_t)(cm_ ? mm_ : 0), src);
                        if (mj == 0) continue;
                        const double vj = lane_f64(valh_, src) * 2.0;      // (exact; 0.0 where the row is not covered at all)
                        acc += (row == r && (uint32_t)(p - aj) < (uint32_t)mj) ? vj : 0.0;
                    }
                }
            }
            return;
        }
        if (PC_CENTER_SKIP & 1) { asm volatile("" : [acc] "+v"(acc) : [cm] "v"(cm_), [val] "v"(valh_)); return; }
        if (nsteps > 12) {
            PC_CENTER_STEPS(PC_CS_Q0 PC_CS_Q1 PC_CS_Q2 PC_CS_Q3);
        } else {
            PC_CENTER_STEPS(PC_CS_Q0);
            if (nsteps > 4) PC_CENTER_STEPS(PC_CS_Q1);
            if (nsteps > 8) PC_CENTER_STEPS(PC_CS_Q2);
        }
    };
    // Long-span reads that start before the near window of a row but may reach into it: they precede every near-window
    // record in the file, so they are replayed first, in list order.  The candidates are the reads whose SPAN covers
    // the chunk -- in a region under many introns nearly all of them put no aligned base on it.  Round 4 walked them 8
    // at a time, one dependent load and 16 replay steps per batch whether or not anything was covered: the chunks under
    // the deepest stacks of spliced reads (7 400 candidates: 930 batches, a microsecond each) WERE the tail of the
    // launch.  Now 64 candidates per trip (one per lane), a vote on which of them can count at all, and only the
    // eighths of the group that hold such a read are handed round (lane permutes) and replayed -- skipping a read that
    // adds +0.0 to every sum leaves every bit as it was.
    const uint32_t long_z = lane_u32(d, kCsLongZ), long_w = lane_u32(d, kCsLongW);
    if (long_w > long_z) {
        const int64_t near_row = (int64_t)rs - W + 1;
        const int64_t near_last = (int64_t)s0 + (kCenterRows - 1) * roww - W + 1;   // the last row's: the furthest any row looks
        for (int64_t base = long_z; base < (int64_t)long_w; base += 64) {
            const int64_t jl = base + lane;
            const bool inl = jl < (int64_t)long_w;
            u32x4 gl = inl ? fv.long_rec[jl] : u32x4{0x7fffffffu, kFlagExcluded << 16, 0u, 0u};
            if ((int64_t)(int32_t)lane_u32(gl.x, 0) >= near_last) break; // sorted by start: the rest is met in the near windows
            const i32x4 rl = inl ? fv.long_runs[jl] : i32x4{0, 0, 0, 0};
            const uint32_t fll = rec_flags(gl.y);
            int nbl = rec_nblk(gl.y), Ll = rec_len(gl.y);
            if (inl && (fll & kFlagWide)) { const u32x2 tv = fv.long_wide[jl]; Ll = (int)tv.x; nbl = (int)tv.y; }   // beyond the 16 / 8-bit fields
            // can candidate `lane` count anywhere in [s0, cend)?  (reads with more than two runs go through their record: kept)
            bool matters = inl && (int64_t)(int32_t)gl.x < near_last && !(fll & kFlagExcluded) && strand_ok(mode, fll & kFlagReverse) && size_ok(mp, Ll);
            if (matters && nbl <= 2) {
                const int lo0 = nib, hi0 = rl.y < Ll - nib ? rl.y : Ll - nib;                         // run 0: read indices [0, len0)
                const int a0 = rl.x + lo0, m0 = hi0 > lo0 ? hi0 - lo0 : 0;
                const int lo1 = rl.y > nib ? rl.y : nib, hi1 = rl.y + rl.w < Ll - nib ? rl.y + rl.w : Ll - nib;   // run 1: [len0, len0 + len1)
                const int a1 = rl.z + (lo1 - rl.y), m1 = (nbl == 2 && hi1 > lo1) ? hi1 - lo1 : 0;
                matters = (m0 > 0 && a0 < cend && a0 + m0 > s0) || (m1 > 0 && a1 < cend && a1 + m1 > s0);
            }
            const unsigned long long vote = __ballot(matters);
            if (vote == 0ull) continue;
#pragma unroll 1
            for (int e8 = 0; e8 < 8; ++e8) {
                if (((vote >> (8 * e8)) & 0xffull) == 0ull) continue;
                // candidate 8 e8 + (li >> 1) of the group to lanes 2k, 2k + 1 of every row (its first / second aligned run)
                const int src = 8 * e8 + (li >> 1);
                u32x4 g;
                g.x = (uint32_t)__shfl((int)gl.x, src, 64); g.y = (uint32_t)__shfl((int)gl.y, src, 64);
                g.z = 0u; g.w = (uint32_t)__shfl((int)gl.w, src, 64);
                i32x4 runs;
                runs.x = __shfl(rl.x, src, 64); runs.y = __shfl(rl.y, src, 64); runs.z = __shfl(rl.z, src, 64); runs.w = __shfl(rl.w, src, 64);
                const int Lg = __shfl(Ll, src, 64), nbk = __shfl(nbl, src, 64);
                const bool in = base + src < (int64_t)long_w;
                const uint32_t fl = rec_flags(g.y);
                const int r = li & 1;
                const bool ok = in && row_live && (int64_t)(int32_t)g.x < near_row && !(fl & kFlagExcluded) && strand_ok(mode, fl & kFlagReverse);
                // (these entries come with their read's header, not from the stream: trimmed by the nibble here)
                const int x = r ? runs.z : runs.x, len = r ? runs.w : runs.y, cum = r ? runs.y : 0;
                const int lo_i = cum > nib ? cum : nib, hi_i = cum + len < Lg - nib ? cum + len : Lg - nib;
                const int a0_ = x + (lo_i - cum), mm_ = hi_i > lo_i ? hi_i - lo_i : 0, mtot = Lg - 2 * nib;
                const int cm_ = (ok && nbk <= 2 && (r == 0 || nbk == 2) && size_ok(mp, Lg)) ? row_mask(a0_, mm_) : 0;
                double valh_ = 0.0;
                if (cm_) valh_ = mtot < 65536 ? invh[mtot] : (1.0 / (double)mtot) * 0.5;   // cm != 0 implies mtot >= m > 0
                replay(a0_, mm_, cm_, valh_, ok && nbk > 2 && r == 0, g.w, 16, true);
            }
        }
    }
    // near windows: row r replays the stream entries [lo, hi) the descriptor names for it
    const uint32_t nmax = lane_u32(d, kCsNmax);
    if (nmax != 0u && !(PC_CENTER_SKIP & 2)) {
        const uint32_t ent0 = lane_u32(d, kCsEnt0), to_end = lane_u32(d, kCsToEnd);
        const uint32_t lo = (uint32_t)__shfl((int)d, kCsLo + row, 64), hi = (uint32_t)__shfl((int)d, kCsHi + row, 64);
        const uint2 *ent_sel = sel == 0 ? cx.ent[0] : (sel == 1 ? cx.ent[1] : cx.ent[2]);   // (selects, not an indexed copy: that would live in scratch)
        const char PC_GLOBAL *eb = MULTI ? (const char PC_GLOBAL *)(cs_stream(cx.files + file, sel) + ent0)
                                         : (const char PC_GLOBAL *)((const u32x2 PC_GLOBAL *)ent_sel + ent0);
        // (loads past a row's end read one of the 64 entries behind the stream's last, which cover nothing; a stream whose
        // end lies beyond a 32-bit byte offset from here clamps to the row's end instead and tests every entry's index)
        const bool far = GENERAL && to_end >= (1u << 28);
        const uint32_t rlo = lo + (uint32_t)li, rhi = hi, dead = far ? rhi : to_end + (uint32_t)li;
        auto fetch = [&](uint32_t base) {
            const uint32_t idx = rlo + base;
            return *(const u32x2 PC_GLOBAL *)(eb + ((idx < rhi ? idx : dead) << 3));
        };
        const bool any_indirect = GENERAL && (MULTI ? cx.files[file].cs_indirect[sel] != 0u : ((cx.indirect >> sel) & 1u) != 0u);   // (uniform: short-read files have none, and never look)
        struct Prepared { int cm; double valh; bool ind; u32x2 r; };
        auto unpack = [&](const u32x2 r, uint32_t base) {
            Prepared pr;
            pr.r = r;
            pr.ind = any_indirect && ((r.y >> 24) & kCsIndirect);     // (an indirect entry carries m = 0)
            pr.cm = row_mask((int32_t)r.x, (int)(r.y & 0xffu));
            if (far && !(rlo + base < rhi)) pr.cm = 0;
            pr.valh = s_valh[(r.y >> 16) & 0xffu];   // by aligned length (an indirect entry reads [0])
            return pr;
        };
        // PC_CENTER2_RING batches in flight, each in a register pair of its own; the batch after the one being replayed is
        // already unpacked.  A batch that starts behind the longest row is not requested at all (uniform test).
        const u32x2 none = {0x7fffffffu, 0u};
        u32x2 q[RING];
#pragma unroll
        for (int k = 0; k < RING; ++k) q[k] = 16u * (uint32_t)k < nmax ? fetch(16u * (uint32_t)k) : none;
        Prepared nxt = unpack(q[0], 0u);
        q[0] = 16u * RING < nmax ? fetch(16u * RING) : none;
        for (uint32_t base = 0; base < nmax;) {
#pragma unroll
            for (int k = 0; k < RING; ++k) {
                const Prepared cur = nxt;
                const int kn = (k + 1) % RING;   // (a constant once the loop is unrolled: q stays in registers)
                nxt = unpack(q[kn], base + 16u);
                q[kn] = base + 16u + 16u * RING < nmax ? fetch(base + 16u + 16u * RING) : none;
                const uint32_t left = nmax - base;
                replay((int32_t)cur.r.x, (int)(cur.r.y & 0xffu), cur.cm, cur.valh, cur.ind, cur.r.x, left >= 16u ? 16 : (int)((left + 3u) & ~3u), GENERAL);
                base += 16u;
                if (base >= nmax) break;
            }
        }
    }
    // the sums, straight into the caller's layout (SegmentChain.get_counts, roitools.pyx:3259-3271; reads-per-million as
    // count / sum * 1e6 in that order, genome_array.py:826-827)
    if (MULTI && !last) return;   // (the next file's reads follow in the same sums)
    if (PC_CENTER_SKIP & 4) { asm volatile("" : : "v"(acc)); return; }
    const double val = cx.norm_on ? acc / cx.norm_sum * 1e6 : acc;
    if ((shape >> 24) & 1u) {   // THE output piece of this chunk, from the descriptor
        const long long out_off = (long long)(((unsigned long long)lane_u32(d, kCsOutHi) << 32) | lane_u32(d, kCsOutLo));
        const int32_t ostart = (int32_t)lane_u32(d, kCsOStart), olen = (int32_t)lane_u32(d, kCsOLen), ostep = (int32_t)lane_u32(d, kCsOStep);
        const uint32_t rel = (uint32_t)(p - ostart);
        if (owns && rel < (uint32_t)olen) cx.out[out_off + (long long)ostep * (long long)rel] = val;
    } else {
        const uint32_t op_begin = lane_u32(d, kCsOpBegin), op_end = lane_u32(d, kCsOpEnd);
        for (uint32_t oi = op_begin; oi < op_end; ++oi) {
            const OutPiece o = cx.opieces[oi];
            if (o.mode != mode) continue;                     // the window's slices of other strand modes
            const uint32_t rel = (uint32_t)(p - o.start);
            if (owns && rel < (uint32_t)o.len) cx.out[o.out_off + (int64_t)o.step * (int64_t)rel] = val;
        }
    }
}

// Dispatch: wave b < n_heavy serves heavy entry b (they start at t = 0); the others serve PC_CENTER_PER_WAVE
// consecutive light entries each, dealt so that the workgroups of one XCD (workgroup b runs on XCD b mod 8) walk ONE
// contiguous eighth of the light list -- neighbouring chunks re-read each other's halo, which then hits the XCD's own L2.
// (the diagnostic instantiations carry their clocks and step counters: compiled for seven waves, they keep out of scratch)
template <bool DBG, bool GENERAL, bool MULTI = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MULTI ? 6 : (DBG ? 7 : 8), 8))) void k_center2(Center2Ctx cx) {
    const uint32_t n_heavy = cx.known ? cx.n_heavy : cx.counters[0], n_light = cx.known ? cx.n_light : cx.counters[1];
    const uint32_t bidx = blockIdx.x;
    const int lane = threadIdx.x & 63;
    constexpr uint32_t K = PC_CENTER_PER_WAVE;
    uint32_t first, count;
    if (bidx < n_heavy) {
        // A heavy entry is ONE long dependent replay (the deepest pile-ups: 15 - 30 k steps), and the launch ends when the
        // last of them does: sharing its SIMD round-robin with seven light waves it advances one step per ~150 cycles
        // (0.95 ms for 15 k steps, the whole tail of the launch).  With a raised priority the arbiter issues it whenever it
        // is ready; the light waves fill the gaps.
        if (PC_CENTER_HEAVY_PRIO) __builtin_amdgcn_s_setprio(PC_CENTER_HEAVY_PRIO);
        first = bidx; count = 1u;
    } else {
        const uint32_t k = bidx - n_heavy, n8 = (n_light + 7u) >> 3;           // entries per eighth of the light list
        const uint32_t x = bidx & 7u, at = (k >> 3) * K;                         // this XCD's eighth, K entries from `at` on
        const uint32_t lo8 = x * n8, hi8 = lo8 + n8 < n_light ? lo8 + n8 : n_light;
        if (lo8 + at >= hi8) return;
        first = n_heavy + lo8 + at;
        count = hi8 - (lo8 + at) < K ? hi8 - (lo8 + at) : K;
    }
    const uint32_t PC_GLOBAL *sw = (const uint32_t PC_GLOBAL *)cx.slots;
    const uint32_t nf = MULTI ? (uint32_t)cx.nfiles : 1u;   // descriptors per entry
    uint32_t d = sw[(size_t)first * nf * 32u + (uint32_t)(lane & 31)];
    unsigned long long *dbg = DBG ? cx.dbg : nullptr;
    const unsigned long long t_begin = dbg ? wall_clock64() : 0ull;
    unsigned long long n_slots = 0;   // PC_CENTER_DEBUG / pc_center_replay_steps: replay steps of this wave
    // half the value of a read by aligned length (k_center_vals), in LDS: one copy per wave
    __shared__ double s_valh[256];
    if (!(PC_CENTER_SKIP & 8)) for (int i = lane; i < 256; i += 64) s_valh[i] = ((const double PC_GLOBAL *)cx.cvalh)[i];
    __builtin_amdgcn_wave_barrier();
    if (MULTI) {
        // several files: the entry's descriptors one after the other into the same sums (file-major, genome_array.py:800-809)
        for (uint32_t c = 0; c < count; ++c) {
            double acc = 0.0;
            for (uint32_t f = 0; f < nf; ++f) {
                uint32_t dn = 0u;
                if (f + 1u < nf) dn = sw[((size_t)(first + c) * nf + f + 1u) * 32u + (uint32_t)(lane & 31)];
                else if (c + 1u < count) dn = sw[(size_t)(first + c + 1u) * nf * 32u + (uint32_t)(lane & 31)];
                center_slot<DBG, GENERAL, PC_CENTER2_RING, true>(cx, d, -1, lane, s_valh, n_slots, acc, (int)f, f + 1u == nf);
                d = dn;
            }
        }
    } else if (!GENERAL && PC_CENTER_HEAVY_RING != PC_CENTER2_RING && bidx < n_heavy) {   // (the short-read instantiation: with the indirect-entry path unrolled into both rings the other one spills)
        // a heavy entry: PC_CENTER_HEAVY_RING batches in flight (four bound a 32 k-step replay by the memory round trip --
        // 64 steps per ~3 600 cycles -- not by the steps)
        double acc = 0.0;
        center_slot<DBG, GENERAL, PC_CENTER_HEAVY_RING>(cx, d, -1, lane, s_valh, n_slots, acc);
    } else {
        for (uint32_t c = 0; c < count; ++c) {
            uint32_t dn = 0u;
            if (c + 1u < count) dn = sw[(size_t)(first + c + 1u) * 32u + (uint32_t)(lane & 31)];   // the next descriptor, in flight while this chunk replays
            double acc = 0.0;
            center_slot<DBG, GENERAL, PC_CENTER2_RING>(cx, d, -1, lane, s_valh, n_slots, acc);
            d = dn;
        }
    }
    if (dbg && lane == 0 && first < cx.dbg_cap) {   // PC_CENTER_DEBUG: heavy entries from the front of the table, light ones from its back (as k_center's list)
        const size_t at = first < n_heavy ? (size_t)first : (size_t)cx.dbg_cap - 1u - (size_t)(first - n_heavy);
        dbg[2 * at] = wall_clock64() - t_begin; dbg[2 * at + 1] = t_begin;
        dbg[2 * (size_t)cx.dbg_cap + at] = n_slots;
    }
}

// ---------------------------------------------------------------- k_coordinates
// SegmentChain._get_position_hash / get_position_list (roitools.pyx:1450-1484, 2059-2080) for a whole
// batch of chains: the genomic coordinate of every element of the plan's output layout (with the
// ascending layout of IntervalTable.position_arrays this is the flat position hash of every chain).
__global__ __launch_bounds__(kWG) void k_coordinates(const GatherSeg *__restrict__ segs,
                                                     const GatherChunk *__restrict__ chunks, int rows, int64_t *out) {
    const GatherChunk gc = chunks[blockIdx.x];
    const GatherSeg sg = segs[gc.seg];
    if (sg.step == 0) return;   // a summed slice has no per-position elements
    const int64_t base = (int64_t)gc.chunk * kGatherChunk;
    const int64_t n = (sg.len - base < kGatherChunk) ? sg.len - base : kGatherChunk;
    for (int r = 0; r < rows; ++r) {
        int64_t *dst = out + sg.out_off + (int64_t)r * sg.row_stride;
        for (int64_t i = threadIdx.x; i < n; i += kWG) dst[(int64_t)sg.step * (base + i)] = sg.start + base + i;
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

// ---------------------------------------------------------------- k_update_flags
// Host-side read filters changed (genome_array.py:697-722, 819-820): rewrite the strand / excluded
// bits of every staged copy of the record headers -- packed record, 4-byte stream word, side lists.
__global__ __launch_bounds__(kWG) void k_update_flags(uint2 *rec, uint32_t *stream, const uint8_t *__restrict__ flags,
                                                      int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kWG + threadIdx.x;
    if (i >= n) return;
    const uint32_t keep = ~((kFlagReverse | kFlagExcluded | kFlagUser) << 16);
    uint2 r = rec[i];
    r.y = (r.y & keep) | (caller_flags(flags[i]) << 16);
    rec[i] = r;
    stream[i] = stream_word(r.x, r.y);
}

// The vectorised read filter on the SAM FLAG word and MAPQ (pc_set_flag_filter): a record stays iff
// (flag & require) == require, (flag & exclude) == 0 and mapq >= min_mapq -- what a filter function such as
// `lambda read: not read.is_secondary and read.mapping_quality >= 10` decides per read on the host
// (genome_array.py:697-722, applied :819-820), here one pass over 3 bytes per record in HBM.  The caller's own
// exclusions (kFlagUser) stay; `enabled` = 0 restores them alone.
// `max_nh` (0: no such test; round 6): keep a read only if it carries an NH:i tag of at most `max_nh` reported alignments
// -- `read.has_tag("NH") and read.get_tag("NH") <= max_nh`, the unique-mapper filter for max_nh = 1; `nh` = the tag's
// value per record, 0 without one.
__global__ __launch_bounds__(kWG) void k_flag_filter(uint2 *rec, uint32_t *stream, const uint16_t *__restrict__ sam_flag,
                                                     const uint8_t *__restrict__ mapq, int64_t n, uint32_t enabled, uint32_t require,
                                                     uint32_t exclude, uint32_t min_mapq, const uint16_t *__restrict__ nh, uint32_t max_nh) {
    const int64_t i = (int64_t)blockIdx.x * kWG + threadIdx.x;
    if (i >= n) return;
    uint2 r = rec[i];
    bool out = ((r.y >> 16) & kFlagUser) != 0u;
    if (enabled) {
        if (sam_flag) {   // (nullptr: an NH filter alone, on a file without the FLAG / MAPQ columns)
            const uint32_t f = sam_flag[i];
            out = out || (f & require) != require || (f & exclude) != 0u || (uint32_t)mapq[i] < min_mapq;
        }
        if (max_nh) { const uint32_t v = nh[i]; out = out || v == 0u || v > max_nh; }
    }
    const uint32_t y = (r.y & ~(kFlagExcluded << 16)) | (out ? kFlagExcluded << 16 : 0u);
    if (y == r.y) return;
    r.y = y;
    rec[i] = r;
    stream[i] = stream_word(r.x, r.y);
}

__global__ __launch_bounds__(kWG) void k_update_run_flags(uint2 *runs, const uint32_t *__restrict__ run_recidx, int64_t n,
                                                          const uint2 *__restrict__ rec) {
    const int64_t j = (int64_t)blockIdx.x * kWG + threadIdx.x;
    if (j >= n) return;
    uint2 r = runs[j];
    const uint32_t fl = (rec[run_recidx[j]].y >> 16) & (kFlagReverse | kFlagExcluded);
    r.y = (r.y & ~((kFlagReverse | kFlagExcluded) << 24)) | (fl << 24);
    runs[j] = r;
}

// ---------------------------------------------------------------- run stream construction (staging time)
// 64-bit sort key of every run: contig << 32 | run start; the payload is permuted with the sorted indices.
__global__ __launch_bounds__(kWG) void k_run_keys(const uint2 *__restrict__ val, const uint32_t *__restrict__ recidx, int64_t n,
                                                  const int64_t *__restrict__ tid_bounds, int ntid, unsigned long long *key,
                                                  uint32_t *order) {
    const int64_t j = (int64_t)blockIdx.x * kWG + threadIdx.x;
    if (j >= n) return;
    const int64_t i = recidx[j];
    int lo = 0, hi = ntid;   // contig of record i
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (tid_bounds[mid + 1] <= i) lo = mid + 1; else hi = mid;
    }
    key[j] = ((unsigned long long)(uint32_t)lo << 32) | val[j].x;
    order[j] = (uint32_t)j;
}

__global__ __launch_bounds__(kWG) void k_run_gather(const uint32_t *__restrict__ order, const uint2 *__restrict__ val_in,
                                                    const uint32_t *__restrict__ idx_in, int64_t n, uint2 *val_out,
                                                    uint32_t *idx_out) {
    const int64_t j = (int64_t)blockIdx.x * kWG + threadIdx.x;
    if (j >= n) return;
    const uint32_t o = order[j];
    val_out[j] = val_in[o];
    idx_out[j] = idx_in[o];
}

// rlin_tab[g] = first sorted run whose (contig, start) is not before bucket g's edge (entry nb of a contig =
// its end), from the sorted 64-bit keys: one bisection per table entry
__global__ __launch_bounds__(kWG) void k_run_lin(const unsigned long long *__restrict__ keys, int64_t n,
                                                 const int64_t *__restrict__ lin_off, int ntid, int64_t nlin, uint32_t *rlin) {
    const int64_t g = (int64_t)blockIdx.x * kWG + threadIdx.x;
    if (g >= nlin) return;
    int lo_t = 0, hi_t = ntid;             // contig of table entry g
    while (lo_t < hi_t) {
        const int mid = (lo_t + hi_t) >> 1;
        if (lin_off[mid + 1] <= g) lo_t = mid + 1; else hi_t = mid;
    }
    const int t = lo_t;
    const int64_t k = g - lin_off[t], nb = lin_off[t + 1] - lin_off[t] - 1;
    const unsigned long long key = k >= nb ? ((unsigned long long)(t + 1) << 32) : (((unsigned long long)t << 32) | (unsigned long long)(k << kLinShift));
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    rlin[g] = (uint32_t)lo;
}

// ---------------------------------------------------------------- side lists, built on the GPU
// The gapped-record, long-span and long-span-outside-the-run-stream lists of a staged file are compactions of its
// records (the class of a record is in its header), so they are made where the records already are: the members are
// selected in record order (stage_kernels.hip.h: k_classify counts them per workgroup, k_side_select writes their record
// indices), k_side_fill writes their entries, an inclusive max-scan of (contig << 32 | end) gives the running maximum
// of the ends per contig, k_list_bounds the per-contig ranges and k_lin_table the linear-index tables.

// aligned runs of the multi-run records as {start, length} pairs, from the caller's two arrays
__global__ __launch_bounds__(kWG) void k_zip_runs(const int32_t *__restrict__ start, const int32_t *__restrict__ len, int64_t n, int2 *blk) {
    const int64_t j = (int64_t)blockIdx.x * kWG + threadIdx.x;
    if (j < n) blk[j] = make_int2(start[j], len[j]);
}

// entry k of a side list = record idx[k]: {pos, header, first run, record}, its first two runs (a single-run record:
// {pos, L}), its contig, (contig << 32 | end) for the running maximum, and -- files with wide records -- the true
// {aligned length, run count}
__global__ __launch_bounds__(kWG) void k_side_fill(const uint32_t *__restrict__ idx, int64_t m, const uint2 *__restrict__ rec,
                                                   const uint32_t *__restrict__ blk_off, const int2 *__restrict__ blk,
                                                   const int64_t *__restrict__ tid_bounds, int ntid,
                                                   const uint32_t *__restrict__ wide_rec, const uint2 *__restrict__ wide_val, int64_t nwide,
                                                   uint4 *out_rec, int4 *out_runs, int32_t *out_tid, unsigned long long *out_key,
                                                   uint2 *out_wide) {
    const int64_t k = (int64_t)blockIdx.x * kWG + threadIdx.x;
    if (k >= m) return;
    const uint32_t i = idx[k];
    const uint2 r = rec[i];
    uint32_t L = r.y & 0xffffu, nb = r.y >> 24;
    if ((r.y >> 16) & kFlagWide) {
        int64_t lo = 0, hi = nwide;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (wide_rec[mid] < i) lo = mid + 1; else hi = mid;
        }
        L = wide_val[lo].x;
        nb = wide_val[lo].y;
    }
    const uint32_t boff = nb >= 2u ? blk_off[i] : 0u;
    int4 runs = make_int4((int32_t)r.x, (int32_t)L, 0, 0);
    int64_t end = (int64_t)(int32_t)r.x + (L > 0u ? (int64_t)L : 1);
    if (nb >= 2u) {
        const int2 b0 = blk[boff], b1 = blk[boff + 1u], bl = blk[boff + nb - 1u];
        runs = make_int4(b0.x, b0.y, b1.x, b1.y);
        end = (int64_t)bl.x + bl.y;
    }
    out_rec[k] = make_uint4(r.x, r.y, boff, i);
    out_runs[k] = runs;
    if (out_tid || out_key) {
        int lo = 0, hi = ntid;   // contig of record i: tid_bounds[t] <= i < tid_bounds[t + 1]
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (tid_bounds[mid + 1] <= (int64_t)i) lo = mid + 1; else hi = mid;
        }
        if (out_tid) out_tid[k] = lo;
        if (out_key) out_key[k] = ((unsigned long long)lo << 32) | (unsigned long long)(uint32_t)end;
    }
    if (out_wide) out_wide[k] = make_uint2(L, nb);
}

// the low half of the scanned keys: the running maximum of the ends inside the entry's contig
__global__ __launch_bounds__(kWG) void k_unpack_pmax(const unsigned long long *__restrict__ key, int64_t m, int32_t *pmax) {
    const int64_t k = (int64_t)blockIdx.x * kWG + threadIdx.x;
    if (k < m) pmax[k] = (int32_t)(uint32_t)key[k];
}

// bounds[t] = entries of the list that belong to contigs before t (the list is in record order)
__global__ __launch_bounds__(kWG) void k_list_bounds(const uint32_t *__restrict__ idx, int64_t m, const int64_t *__restrict__ tid_bounds,
                                                     int ntid, int64_t *bounds) {
    const int t = (int)(blockIdx.x * kWG + threadIdx.x);
    if (t > ntid) return;
    const int64_t first = tid_bounds[t];
    int64_t lo = 0, hi = m;
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if ((int64_t)idx[mid] < first) lo = mid + 1; else hi = mid;
    }
    bounds[t] = lo;
}

// Linear index of a list: tab[g] = first entry of bucket g's contig whose key is not before the bucket's edge (the
// last entry of a contig's table = the end of its range).  KEY 0: the start of a record (rec[].x); 1: the start of a
// side-list entry (uint4 .x); 2: a running maximum of ends -- an entry counts as "before" while its maximum is <= edge.
template <int KEY>
__global__ __launch_bounds__(kWG) void k_lin_table(const void *__restrict__ keys, const int64_t *__restrict__ bounds,
                                                   const int64_t *__restrict__ lin_off, int ntid, int64_t nlin, uint32_t *tab) {
    const int64_t g = (int64_t)blockIdx.x * kWG + threadIdx.x;
    if (g >= nlin) return;
    int lo_t = 0, hi_t = ntid;             // contig of table entry g
    while (lo_t < hi_t) {
        const int mid = (lo_t + hi_t) >> 1;
        if (lin_off[mid + 1] <= g) lo_t = mid + 1; else hi_t = mid;
    }
    const int t = lo_t;
    const int64_t k = g - lin_off[t], nb = lin_off[t + 1] - lin_off[t] - 1;
    int64_t lo = bounds[t], hi = bounds[t + 1];
    if (k >= nb) { tab[g] = (uint32_t)hi; return; }
    const int64_t edge = k << kLinShift;
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        bool before;
        if (KEY == 0) before = (int64_t)(int32_t)((const uint2 *)keys)[mid].x < edge;
        else if (KEY == 1) before = (int64_t)(int32_t)((const uint4 *)keys)[mid].x < edge;
        else before = (int64_t)((const int32_t *)keys)[mid] <= edge;
        if (before) lo = mid + 1; else hi = mid;
    }
    tab[g] = (uint32_t)lo;
}

__global__ __launch_bounds__(kWG) void k_update_side_flags(uint4 *list, int64_t n, const uint2 *__restrict__ rec) {
    const int64_t j = (int64_t)blockIdx.x * kWG + threadIdx.x;
    if (j >= n) return;
    uint4 g = list[j];
    g.y = rec[g.w].y; // the side lists carry a copy of the header
    list[j] = g;
}

// ---------------------------------------------------------------- run-length encoding
// Export side (genome_array.py:1041-1111, to_bedgraph / to_variable_step): the per-position
// vector of a whole chromosome is reduced on the GPU to its runs -- a run starts at element 0,
// where the value changes, and at every multiple of `period` (the scripts cut their runs at
// window borders; period 1 lists every element) -- so that only the runs cross PCIe.
// 64-bit patterns are compared (int64 counts or float64 values).  Two passes over `kRleChunk`
// elements per workgroup: count the heads, scan the counts, write {start, value}.
constexpr int kRleChunk = 2048;
__device__ __forceinline__ bool rle_head(const unsigned long long PC_GLOBAL *v, int64_t i, int64_t period) {
    return i == 0 || v[i] != v[i - 1] || (period > 0 && i % period == 0);
}

__global__ __launch_bounds__(kWG) void k_rle_count(const unsigned long long *__restrict__ v_, int64_t n, int64_t period,
                                                   uint32_t *wg_count) {
    const unsigned long long PC_GLOBAL *v = (const unsigned long long PC_GLOBAL *)v_;
    __shared__ uint32_t s_wave[kWG / 64];
    const int64_t base = (int64_t)blockIdx.x * kRleChunk;
    uint32_t c = 0;
    for (int k = 0; k < kRleChunk / kWG; ++k) {
        const int64_t i = base + k * kWG + threadIdx.x;
        if (i < n && rle_head(v, i, period)) ++c;
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < kWG / 64; ++w) t += s_wave[w];
        wg_count[blockIdx.x] = t;
    }
}

// exclusive scan of the per-workgroup counts by ONE workgroup (a few thousand entries per Mb)
__global__ __launch_bounds__(kWG) void k_rle_scan(const uint32_t *__restrict__ wg_count, int64_t nwg, int64_t *wg_base,
                                                  int64_t *total) {
    __shared__ int64_t s_part[kWG];
    const int64_t per = (nwg + kWG - 1) / kWG;
    const int64_t b = (int64_t)threadIdx.x * per, e = b + per < nwg ? b + per : nwg;
    int64_t sum = 0;
    for (int64_t i = b; i < e; ++i) sum += wg_count[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t run = 0;
        for (int t = 0; t < kWG; ++t) { const int64_t x = s_part[t]; s_part[t] = run; run += x; }
        *total = run;
    }
    __syncthreads();
    int64_t run = s_part[threadIdx.x];
    for (int64_t i = b; i < e; ++i) { wg_base[i] = run; run += wg_count[i]; }
}

__global__ __launch_bounds__(kWG) void k_rle_write(const unsigned long long *__restrict__ v_, int64_t n, int64_t period,
                                                   const int64_t *__restrict__ wg_base, int64_t *starts,
                                                   unsigned long long *values) {
    const unsigned long long PC_GLOBAL *v = (const unsigned long long PC_GLOBAL *)v_;
    __shared__ uint32_t s_wave[kWG / 64];
    const int64_t base = (int64_t)blockIdx.x * kRleChunk;
    int64_t out = wg_base[blockIdx.x];
    for (int k = 0; k < kRleChunk / kWG; ++k) { // element order == run order
        const int64_t i = base + k * kWG + threadIdx.x;
        const bool head = i < n && rle_head(v, i, period);
        const unsigned long long m = __ballot(head);
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 0) s_wave[wv] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0, tot = 0;
        for (int w = 0; w < kWG / 64; ++w) { const uint32_t t = s_wave[w]; if (w < wv) before += t; tot += t; }
        if (head) {
            const int64_t slot = out + before + __popcll(m & ((1ull << lane) - 1ull));
            starts[slot] = i;
            values[slot] = v[i];
        }
        out += tot;
        __syncthreads();
    }
}

// ---------------------------------------------------------------- streaming probes
// Measurement helpers (pc_stream_probe): what this GPU sustains for the two access patterns of the
// tile kernel -- 16-byte-per-lane contiguous loads (one contiguous chunk per workgroup, four loads
// in flight per lane) and 8-byte-per-lane contiguous stores.  The measured read rate is the second
// roofline denominator of bench.py; the store kernel is also the known byte count on
// which the WRITE_SIZE counter is calibrated (profiles/traffic.json).
constexpr int kProbeChunk = 16384;   // 16-byte vectors per workgroup (256 KiB)
__global__ __launch_bounds__(kWG) void k_probe_read(const u32x4 *__restrict__ src_, int64_t nvec, uint32_t *sink) {
    const u32x4 PC_GLOBAL *src = (const u32x4 PC_GLOBAL *)src_;
    const int64_t lo = (int64_t)blockIdx.x * kProbeChunk;
    const int64_t hi = lo + kProbeChunk < nvec ? lo + kProbeChunk : nvec;
    uint32_t acc = 0;
    for (int64_t b = lo + threadIdx.x; b < hi; b += 4 * kWG) {
        u32x4 r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = b + u * kWG < hi ? src[b + u * kWG] : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= r[u].x ^ r[u].y ^ r[u].z ^ r[u].w;
    }
    if (acc == 0x9e3779b9u) *sink = acc;   // never true for the probe's fill pattern; keeps the loads alive
}

__global__ __launch_bounds__(kWG) void k_probe_write(unsigned long long *dst, int64_t n) {
    const int64_t lo = (int64_t)blockIdx.x * (2 * kProbeChunk);
    const int64_t hi = lo + 2 * kProbeChunk < n ? lo + 2 * kProbeChunk : n;
    for (int64_t i = lo + threadIdx.x; i < hi; i += kWG) dst[i] = (unsigned long long)i;
}

// ---------------------------------------------------------------- k_mapped_reads
// reads_out of the map functions (genome_array.py:800-823): is record i of the file among the reads the reference's
// map function appends for the segment [start, end) -- fetched (htslib overlap: pos < end, endpos > start), on the
// segment's strand, past the filters, and mapped by the rule to a position inside the segment (center rule: every
// fetched read with a positive map length, map_factories.pyx:249-256: appended even if nothing landed)?
__device__ __forceinline__ bool read_is_mapped(const GFile &fv, const MapParams &mp, int64_t i, int64_t start, int64_t end, int mode,
                                               bool strand_filter) {
    const u32x2 r = fv.rec[i];
    const uint32_t meta = r.y;
    const uint32_t fl = rec_flags(meta);
    const int32_t pos = (int32_t)r.x;
    int L, nb;
    rec_true(fv, i, meta, L, nb);
    const bool rev = fl & kFlagReverse;
    const bool fetched = (int64_t)pos < end && (int64_t)rec_end(fv, i, pos, meta) > start;
    if (!(fetched && !(fl & kFlagExcluded) && (!strand_filter || strand_ok(mode, rev)) && size_ok(mp, L))) return false;
    if (mp.kind == 2) return (L - 2 * mp.param) > 0;
    int row;
    const int k = map_kleft_dyn(mp, L, mode == 1 || mode == 3, row);
    if (k < 0) return false;
    const int64_t p = nb >= 2 ? walk_runs(fv, i, nb, k) : pos + k;
    return p >= start && p < end;
}

// ONE segment: a mask over the record range [rec_lo, rec_hi)
__global__ __launch_bounds__(kWG) void k_mapped_reads(FileView fview, MapParams mp, int64_t rec_lo, int64_t rec_hi,
                                                      int64_t start, int64_t end, int mode, bool strand_filter,
                                                      uint8_t *mask) {
    const GFile fv = gfile(fview);
    int64_t i = rec_lo + (int64_t)blockIdx.x * kWG + threadIdx.x;
    if (i >= rec_hi) return;
    mask[i - rec_lo] = read_is_mapped(fv, mp, i, start, end, mode, strand_filter) ? 1 : 0;
}

// EVERY segment of a batch at once (get_reads for thousands of regions: bin/psite.py:182, bin/phase_by_size.py:187
// loop over it): one workgroup per (segment, file) walks the records that can overlap the segment -- those that start
// in [start - longest span of the file, end) -- in file order.  FILL = false counts the mapped ones, FILL = true
// writes their record indices at the scanned offsets, in order (a CSR over (segment, file)).
struct BatchSeg {
    int64_t start, end;
    int32_t tid;
    int32_t mode;       // strand mode (bit 8: no strand filter)
};

template <bool FILL>
__global__ __launch_bounds__(kWG) void k_mapped_reads_batch(const BatchSeg *__restrict__ segs, int64_t nseg, const FileView *__restrict__ files,
                                                            int nfiles, const int64_t *__restrict__ max_span, MapParams mp,
                                                            unsigned long long *counts, const unsigned long long *__restrict__ offsets,
                                                            uint32_t *rec_out) {
    __shared__ uint32_t s_wave[kWG / 64];
    const int64_t sf = blockIdx.x;
    const int64_t s = sf / nfiles;
    const int f = (int)(sf % nfiles);
    if (s >= nseg) return;
    const BatchSeg sg = segs[s];
    const GFile fv = gfile(files[f]);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long n = 0;
    if (sg.tid >= 0 && sg.end >= sg.start) {   // (an empty segment still fetches the reads that span it: pos < end, endpos > start)
        const int64_t q0 = fv.lin_off[sg.tid], nb = fv.lin_off[sg.tid + 1] - q0 - 1;
        const int64_t t0 = fv.tid_bounds[sg.tid], t1 = fv.tid_bounds[sg.tid + 1];
        int64_t lo = t0, hi = t1;
        if (nb > 0) {
            lo = bucket_lower_bound(fv, q0, nb, sg.start - max_span[f] + 1);
            hi = bucket_lower_bound(fv, q0, nb, sg.end);
        } else {
            lo = hi = t0;
        }
        const int mode = sg.mode & 3;
        const bool strand_filter = !(sg.mode & 0x100);
        unsigned long long out = FILL ? offsets[sf] : 0ull;
        for (int64_t base = lo; base < hi; base += kWG) {
            const int64_t i = base + threadIdx.x;
            const bool hit = i < hi && read_is_mapped(fv, mp, i, sg.start, sg.end, mode, strand_filter);
            const unsigned long long m = __ballot(hit);
            if (!FILL) {
                n += (lane == 0) ? (unsigned long long)__popcll(m) : 0ull;
            } else {
                if (lane == 0) s_wave[wv] = (uint32_t)__popcll(m);
                __syncthreads();
                uint32_t before = 0, tot = 0;
                for (int w = 0; w < kWG / 64; ++w) { const uint32_t t = s_wave[w]; if (w < wv) before += t; tot += t; }
                if (hit) rec_out[out + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)i;
                out += tot;
                __syncthreads();
            }
        }
    }
    if (!FILL) {
        // sum over the workgroup's four waves (lane 0 of each holds its wave's count)
        if (lane == 0) s_wave[wv] = (uint32_t)n;    // (a (segment, file) pair holds fewer than 2^32 records)
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long t = 0;
            for (int w = 0; w < kWG / 64; ++w) t += s_wave[w];
            counts[sf] = t;
        }
    }
}

// ---------------------------------------------------------------- k_gather_records / k_gather_runs
// Read objects for files whose records live in HBM only (`BAMGenomeArray(path, keep_reads=False)`; the reference hands
// pysam reads to its callers, genome_array.py:834-859): the header fields of the records `idx` names -- reference id,
// first aligned position, aligned length, strand, run count, FLAG / MAPQ when the file carries them -- and then their
// aligned runs (a record with one run is the implicit run [pos, pos + L)).
__global__ __launch_bounds__(kWG) void k_gather_records(FileView fview, int ntid, const int64_t *__restrict__ idx, int64_t n,
                                                        const uint16_t *__restrict__ sam_flag, const uint8_t *__restrict__ sam_mapq,
                                                        int32_t *tid, int32_t *pos, int32_t *alen, uint8_t *reverse, int32_t *nblk,
                                                        uint16_t *flag16, uint8_t *mapq) {
    const int64_t k = (int64_t)blockIdx.x * kWG + threadIdx.x;
    if (k >= n) return;
    const GFile fv = gfile(fview);
    const int64_t i = idx[k];
    const u32x2 r = fv.rec[i];
    int L, nb;
    rec_true(fv, i, r.y, L, nb);
    int lo = 0, hi = ntid;   // contig of record i
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (fv.tid_bounds[mid + 1] <= i) lo = mid + 1; else hi = mid;
    }
    tid[k] = lo; pos[k] = (int32_t)r.x; alen[k] = L; nblk[k] = nb;
    reverse[k] = (uint8_t)(rec_flags(r.y) & kFlagReverse);
    if (flag16) flag16[k] = sam_flag ? sam_flag[i] : (uint16_t)((rec_flags(r.y) & kFlagReverse) ? 0x10 : 0);
    if (mapq) mapq[k] = sam_mapq ? sam_mapq[i] : (uint8_t)255;
}

// run_at[k] = first slot of record k's runs in the output (exclusive sum of max(nblk, 1) over the request, 0 runs for L = 0)
__global__ __launch_bounds__(kWG) void k_gather_runs(FileView fview, const int64_t *__restrict__ idx, int64_t n, const int64_t *__restrict__ run_at,
                                                     int32_t *start, int32_t *len) {
    const int64_t k = (int64_t)blockIdx.x * kWG + threadIdx.x;
    if (k >= n) return;
    const GFile fv = gfile(fview);
    const int64_t i = idx[k];
    const u32x2 r = fv.rec[i];
    int L, nb;
    rec_true(fv, i, r.y, L, nb);
    const int64_t at = run_at[k];
    if (nb >= 2) {
        const uint32_t off = fv.blk_off[i];
        for (int q = 0; q < nb; ++q) { const i32x2 b = fv.blk[off + q]; start[at + q] = b.x; len[at + q] = b.y; }
    } else if (L > 0) { start[at] = (int32_t)r.x; len[at] = L; }
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
    int L, nb_unused;
    rec_true(fv, i, meta, L, nb_unused);
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
    u.len = L;
    u.rec = (uint32_t)i;
    list[slot] = u;
}

} // namespace pc
