/*
 * plastid_counts.h -- C ABI of the MI355X per-position read-counting engine.
 *
 * This is the drop-in boundary for ONE hot path of plastid (reference paths are
 * relative to the reference checkout):
 *
 *   plastid/genomics/genome_array.py:760-832   BAMGenomeArray.get_reads_and_counts
 *   plastid/genomics/genome_array.py:861-928   BAMGenomeArray.__getitem__ / get
 *   plastid/genomics/map_factories.pyx:167-839 the five BAM mapping functions + SizeFilterFactory
 *   plastid/genomics/roitools.pyx:3221-3315    SegmentChain.get_counts / get_masked_counts
 *
 * The reference has no FFI on this path (it is Cython calling pysam); the entry
 * points below are what a binding for the path would bind.  Each one cites the
 * reference interface it replaces.  INTEGRATION.md shows the reference-side
 * ctypes stub.
 *
 * Conventions
 *   - every function returns 0 on success, a negative PC_ERR_* code on failure;
 *     pc_last_error() returns a message for the calling thread's last failure;
 *   - no exceptions cross the ABI, no torch types, plain pointers and sizes;
 *   - host buffers are caller-owned, C-contiguous; device buffers are
 *     engine-owned behind the opaque handles;
 *   - one engine per GPU; an engine and its plans are not thread-safe;
 *   - there is NO CPU fallback: without a usable HIP device pc_create() fails.
 */
#ifndef PLASTID_COUNTS_H
#define PLASTID_COUNTS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever entry points are added or the meaning of an argument changes (6: round 6).  A binding checks it
 * BEFORE it resolves any other symbol: a stale library then fails with a version message, not with a missing symbol. */
#define PC_ABI_VERSION 6

typedef struct pc_engine pc_engine;
typedef struct pc_plan pc_plan;

/* error codes */
#define PC_OK 0
#define PC_ERR_ARG (-1)        /* invalid argument (ValueError in the Python shim)   */
#define PC_ERR_HIP (-2)        /* HIP runtime failure                                */
#define PC_ERR_NOMEM (-3)
#define PC_ERR_UNSORTED (-4)   /* records not coordinate sorted (pysam.fetch raises) */
#define PC_ERR_STATE (-5)      /* e.g. counting before alignments / mapping are set  */

/* mapping rules: plastid/genomics/map_factories.pyx */
#define PC_MAP_FIVE 0    /* FivePrimeMapFactory                    :278-374 */
#define PC_MAP_THREE 1   /* ThreePrimeMapFactory                   :377-474 */
#define PC_MAP_CENTER 2  /* CenterMapFactory                       :167-276 */
#define PC_MAP_VAR5 3    /* VariableFivePrimeMapFactory            :477-650 */
#define PC_MAP_STRAT5 4  /* StratifiedVariableFivePrimeMapFactory  :653-791 */

/* strand codes: plastid/genomics/c_common.pxd:1-6 */
#define PC_STRAND_UNDEF 0
#define PC_STRAND_FWD 1
#define PC_STRAND_REV 2
#define PC_STRAND_UNS 3
/* OR-ed into a segment's strand byte: do not strand-filter reads (semantics of
 * calling a map factory directly on a read list, map_factories.pyx:308, instead
 * of going through BAMGenomeArray, genome_array.py:812-815) */
#define PC_STRAND_NOFILTER 0x10

/* record flag bits (the `flags` array of pc_add_alignment_file) */
#define PC_FLAG_REVERSE 0x01   /* pysam AlignedSegment.is_reverse                       */
#define PC_FLAG_EXCLUDED 0x80  /* dropped by a host-side read filter (genome_array.py:819-820) */

#define PC_OUT_INT64 0
#define PC_OUT_FLOAT64 1

#define PC_OFFSET_TABLE_LEN 10000 /* map_factories.pxd:10-12 */
#define PC_MAX_ALIGNED_LEN 65535 /* of the packed 16-bit field; longer reads: pc_add_alignment_file_wide */

const char *pc_last_error(void);
int pc_abi_version(void);
/* number of HIP devices visible; never initialises more than the runtime needs */
int pc_device_count(void);

/* ---- engine ------------------------------------------------------------- */
int pc_create(int device, pc_engine **out);
/* Every plan of the engine must have been destroyed first (plans return their device blocks to it). */
int pc_destroy(pc_engine *e);
/* A destroyed engine leaves its idle device blocks to the process (up to PC_POOL_RESERVOIR_GB per device, default 4; the
 * last engine of a device to go trims what is kept to that default) for the next engine on the same device --
 * BAMGenomeArray objects come and go, and allocating tens of GB again can take seconds.
 * This returns them to the driver (no reference counterpart; for callers that share the GPU with other libraries). */
int pc_release_cached_memory(int device);
/* Page-locked host memory for count vectors that are read back repeatedly (pc_read_counts into such a buffer is one DMA
 * at the rate of the link, no staging through the transfer ring): `bytes` of it, touched by the calling thread's NUMA
 * policy.  What the reference returns from get_counts is an ordinary numpy array (roitools.pyx:3259-3271); this is an
 * allocator a caller MAY use for the array it hands to pc_read_counts -- no reference counterpart. */
int pc_host_alloc(pc_engine *e, uint64_t bytes, void **out);
int pc_host_free(pc_engine *e, void *p);
/* The PC_* tuning/diagnostic environment knobs are read once, by pc_create;
 * this re-reads them (tests and experiments only -- no reference counterpart). */
int pc_reload_knobs(pc_engine *e);

/* ---- alignments: replaces pysam AlignmentFile.fetch + AlignedSegment.positions /
 * .is_reverse as consumed at genome_array.py:800-815 and map_factories.pyx:243,
 * 349, 448, 629, 769, 838.  One call per BAM file, in the order the files were
 * given to BAMGenomeArray (file-major order matters, genome_array.py:800-809).
 *
 *   n          records, sorted by (tid, pos), ties in file order; at most 2^31 - 2 per file (and
 *              2^31 - 2 runs of multi-run reads): record indices and window bins are 32-bit (PC_ERR_ARG beyond that --
 *              split larger inputs into several files, which are counted as one)
 *   tid,pos    reference index (0 <= tid < ntid) and leftmost aligned coordinate
 *   alen       L = len(read.positions): aligned reference positions (M/=/X)
 *   flags      PC_FLAG_* bits
 *   nblk       number of maximal runs of contiguous aligned positions (0 iff L==0)
 *   nrun,blk_* runs (start,len) of every record with nblk >= 2, record after
 *              record; a record with nblk == 1 is the implicit run [pos, pos+L)
 *
 * The arrays are the caller's (pageable or page-locked host memory) and are only read: they cross PCIe as they are
 * and are validated on the GPU.  A file that breaks the contract is refused (PC_ERR_ARG; PC_ERR_UNSORTED for records
 * out of (tid, pos) order) with the defect of the LOWEST record index in pc_last_error(), and nothing is staged.
 */
int pc_clear_alignments(pc_engine *e);
int pc_add_alignment_file(pc_engine *e, int64_t n, int32_t ntid, const int32_t *tid,
                          const int32_t *pos, const uint16_t *alen, const uint8_t *flags,
                          const uint8_t *nblk, int64_t nrun, const int32_t *blk_start,
                          const int32_t *blk_len);
/* The same with WIDE records: reads whose aligned length exceeds 65 535 or that have more than 255 aligned runs (long
 * reads; the reference has no such limit -- read.positions is a Python list, map_factories.pyx:243, 349).  Such a
 * record carries the markers alen = 65535 AND nblk = 255 in the packed arrays and its true values in the side arrays:
 *   n_wide                 number of wide records
 *   wide_idx               their record indices, ascending
 *   wide_alen, wide_nblk   true L and run count (int32); runs in blk_* as for every record with >= 2 runs */
int pc_add_alignment_file_wide(pc_engine *e, int64_t n, int32_t ntid, const int32_t *tid,
                               const int32_t *pos, const uint16_t *alen, const uint8_t *flags,
                               const uint8_t *nblk, int64_t nrun, const int32_t *blk_start,
                               const int32_t *blk_len, int64_t n_wide, const int64_t *wide_idx,
                               const int32_t *wide_alen, const int32_t *wide_nblk);
/* replace the flags of one staged file (host-side filters changed) */
int pc_update_flags(pc_engine *e, int file, int64_t n, const uint8_t *flags);
int pc_num_files(pc_engine *e);
int64_t pc_num_records(pc_engine *e, int file);
/* Read objects of a staged file back (the reference hands pysam reads to its callers: get_reads / reads_out,
 * genome_array.py:834-859, and to filter functions, :697-722) -- for files whose records never visited the host
 * (pc_add_alignment_bam[_path | _span]).  pc_read_records: per requested record index its reference id, first aligned
 * position, aligned length L = len(read.positions), strand (1: reverse), number of aligned runs, and -- pointers may be
 * NULL -- the SAM FLAG word and MAPQ (from the strand alone / 255 when the file carries none).  pc_read_record_runs: the
 * aligned runs of the same records, record k's from slot run_at[k] on (the caller's exclusive sum of its run counts;
 * nruns = their total): read.positions is their concatenation. */
int pc_read_records(pc_engine *e, int file, int64_t n, const int64_t *idx, int32_t *tid, int32_t *pos, int32_t *alen, uint8_t *reverse,
                    int32_t *nblk, uint16_t *flag16, uint8_t *mapq);
int pc_read_record_runs(pc_engine *e, int file, int64_t n, const int64_t *idx, const int64_t *run_at, int64_t nruns, int32_t *start,
                        int32_t *len);

/* ---- read filters on the SAM FLAG word and MAPQ.  The reference's filter contract is "a function of the
 * pysam.AlignedSegment" (BAMGenomeArray.add_filter, plastid/genomics/genome_array.py:697-722; every filter is called on
 * every fetched read, :819-820).  Filters of the common kind -- `not read.is_secondary`, `not read.is_duplicate`,
 * `read.mapping_quality >= 10`, `read.is_proper_pair` -- depend on two fields of the BAM record only (FLAG: u16 at
 * byte 14, MAPQ: u8 at byte 9 of the record body, kent/src/htslib/sam.c bam_read1; bit names htslib/sam.h:110-132):
 *   pc_set_alignment_sam   hands the engine those two columns of one staged file (3 bytes per record; files staged by
 *                          pc_add_alignment_bam[_path] carry them already);
 *   pc_set_flag_filter     a record is kept iff (flag & require) == require && (flag & exclude) == 0 && mapq >= min_mapq.
 *                          Evaluated on the GPU in one pass per file and folded into the exclusion bit the counting
 *                          kernels honour, on top of the caller's own PC_FLAG_EXCLUDED verdicts; enabled = 0 lifts it.
 *                          PC_ERR_STATE if a staged file has records but no FLAG / MAPQ columns (also from pc_count,
 *                          for a file staged after the filter was set). */
int pc_set_alignment_sam(pc_engine *e, int file, int64_t n, const uint16_t *flag, const uint8_t *mapq);
int pc_set_flag_filter(pc_engine *e, int enabled, uint32_t require, uint32_t exclude, int min_mapq);
/* The same for the NH:i tag (number of reported alignments of the query): `lambda read: read.get_tag("NH") == 1` is the
 * usual unique-mapper filter of the reference's users (genome_array.py:697-722, applied :819-820).
 *   pc_set_alignment_nh    hands the engine the tag's value per record of one staged file (uint16, clamped to 65 535;
 *                          0: the record has no NH tag); files staged by pc_add_alignment_bam* carry it already.
 *   pc_set_nh_filter       max_nh > 0: a record is kept iff it has the tag and its value is <= max_nh; 0 lifts the test.
 *                          Combines with pc_set_flag_filter and the caller's own exclusions (all must keep the read). */
int pc_set_alignment_nh(pc_engine *e, int file, int64_t n, const uint16_t *nh);
int pc_set_nh_filter(pc_engine *e, int max_nh);

/* ---- mapping rule: replaces BAMGenomeArray.set_mapping (genome_array.py:935-963)
 * with one of the five factories.  `param` = offset (FIVE/THREE) or nibble
 * (CENTER).  fw/rc = forward_offsets/reverse_offsets[table_len] built on the
 * host by the restated __cinit__ rules (map_factories.pyx:494-543); entries are
 * -1 (no usable offset) or 0 <= off < L.  min_len/max_len: STRAT5 rows. */
int pc_set_mapping(pc_engine *e, int kind, int param, const int32_t *fw, const int32_t *rc,
                   int table_len, int min_len, int max_len);
/* SizeFilterFactory(min,max) (map_factories.pyx:794-839); max == -1: no maximum */
int pc_set_size_filter(pc_engine *e, int enabled, int min_len, int max_len);
/* set_normalize/set_sum (genome_array.py:482-520): out = count / sum * 1e6 */
int pc_set_normalize(pc_engine *e, int enabled, double sum);
int pc_mapping_rows(pc_engine *e);

/* ---- query plan: a batch of GenomicSegments and where their count vectors go.
 * Replaces the per-segment loop of SegmentChain.get_counts (roitools.pyx:3259-3271)
 * and BAMGenomeArray.get (genome_array.py:891-928).  For segment s, genomic
 * position start[s]+i (0 <= i < end[s]-start[s]), row r is written to
 *     out[out_off[s] + out_step[s]*i + r*row_stride[s]]
 * so a '-' chain is laid out 5'->3' by giving its segments out_step = -1
 * (roitools.pyx:3270-3271, genome_array.py:829-830).  out_step = 0 SUMS the slice instead:
 *     out[out_off[s] + r*row_stride[s]] += sum_i count(start[s]+i, r)
 * (fused region statistics, numpy.nansum(chain.get_masked_counts(ga)) of
 * bin/counts_in_region.py:120; integer rules without normalisation only).  tid < 0 or >= ntid means
 * "chromosome not in the array": the slice is zero (genome_array.py:795-798).
 * A plan depends only on the intervals; it can be reused across alignment sets,
 * mapping rules and normalisation settings with the same number of rows.
 */
int pc_plan_create(pc_engine *e, int64_t nseg, const int32_t *tid, const int64_t *start,
                   const int64_t *end, const uint8_t *strand, const int64_t *out_off,
                   const int8_t *out_step, const int64_t *row_stride, int64_t out_elems, int rows,
                   pc_plan **out);
int pc_plan_destroy(pc_plan *p);
/* SegmentChain.get_position_list / _get_position_hash (roitools.pyx:1450-1484, 2059-2080) for the whole
 * batch: host_out[k] = genomic coordinate of output element k of the plan's layout (-1 where no segment
 * writes; every row of a multi-row layout gets the coordinates; summed slices have none). */
int pc_plan_coordinates(pc_engine *e, pc_plan *p, int64_t *host_out, int64_t out_elems);
int64_t pc_plan_positions(pc_plan *p); /* distinct (strand-mode, position) pairs counted */
int64_t pc_plan_tiles(pc_plan *p);
/* The plan's tables as pc_plan_create built them (tests compare the GPU builder of large annotations with the host
 * builder, table by table).  which: 0 tiles (32-byte records), 1 island pieces (24), 2 output pieces (40), 3 per-segment
 * gather records (64), 4 scalars as int64[12] = {window size, strand-mode mask, modes per window (max), histogram
 * positions, covered output elements, has summed slices, output needs zeroing, tiles, pieces, output pieces, built on
 * the GPU, segments}.  Copies min(cap_bytes, table bytes) to `buf`; *bytes = table bytes. */
int pc_plan_table(pc_plan *p, int which, void *buf, int64_t cap_bytes, int64_t *bytes);

/* ---- counting: map_fn(list(reads), roi) for every segment of the plan
 * (genome_array.py:823), then normalisation/strand layout.  pc_count launches
 * asynchronously on the engine's stream; results stay in HBM until read. */
int pc_count(pc_engine *e, pc_plan *p, int out_dtype);
int pc_sync(pc_engine *e);
/* ONE segment in ONE call: `ga[segment]` / `BAMGenomeArray.get(segment, roi_order)` (genome_array.py:861-928) -- what the
 * reference's scripts do region by region (bin/psite.py:181-192) -- without a plan object: the window travels in the
 * kernel's arguments, the counts come back through page-locked memory the kernel writes itself.  host_out: end - start
 * elements of int64 / float64 (reads per million when normalisation is on), reversed when `reverse_out` (roi_order on
 * a '-' segment, :829-830).  PC_ERR_STATE when the query does not qualify -- several staged files, the center or the
 * stratified rule, a segment longer than 4 096 positions -- and the caller takes pc_plan_create; the DataWarning of the
 * map functions is not reported here (a caller that needs it asks pc_warn_flags through a plan). */
int pc_query_segment(pc_engine *e, int32_t tid, int64_t start, int64_t end, uint8_t strand, int reverse_out, int out_dtype,
                     void *host_out);
int pc_read_counts(pc_engine *e, pc_plan *p, void *host_out, int64_t out_elems);
/* device pointer/stream of the last pc_count output (for zero-copy consumers) */
void *pc_counts_device_ptr(pc_plan *p);
void *pc_stream(pc_engine *e);

/* ---- export: run-length encoding of the last pc_count output on the GPU, so that only the runs
 * cross PCIe (BAMGenomeArray.to_bedgraph / to_variable_step, genome_array.py:990-1111).  A run
 * starts at element 0, wherever the 8-byte value differs from its predecessor, and at every multiple
 * of `period` elements (0: no such cut; the reference cuts its runs at window borders; 1 lists
 * every element).  pc_read_rle returns, for run k, its first element and its value (int64 or
 * float64 as counted); run k ends where run k+1 starts, the last one at out_elems. */
int pc_rle(pc_engine *e, pc_plan *p, int64_t period, int64_t *n_runs);
int pc_read_rle(pc_engine *e, pc_plan *p, int64_t *starts, void *values, int64_t n_runs);

/* per-segment "the reference would emit its DataWarning" flags
 * (map_factories.pyx:258-263, 360-365, 459-464, 643-648) for the last pc_count */
int pc_warn_flags(pc_engine *e, pc_plan *p, uint8_t *flags);
/* the same, plus (last_len, may be NULL) the aligned length of the LAST such read in fetch order per
 * flagged segment, -1 otherwise: the length VariableFivePrimeMapFactory names in its warning
 * (`no_offset_length`, map_factories.pyx:633-648) */
int pc_warn_details(pc_engine *e, pc_plan *p, uint8_t *flags, int32_t *last_len);

/* Sum over all output elements of the last pc_count, left in HBM for an RCCL
 * all-reduce (int64 for PC_OUT_INT64, fixed-order float64 otherwise);
 * host_out8 (8 bytes) may be NULL. */
int pc_total(pc_engine *e, pc_plan *p, void *host_out8);
void *pc_total_device_ptr(pc_plan *p);

/* reads_out of map_fn for ONE segment over records [rec_lo, rec_hi) of `file`
 * (genome_array.py:800-823): mask[i - rec_lo] = 1 iff the record is fetched
 * (htslib overlap), passes strand + filters and is mapped into the segment. */
int pc_mapped_reads(pc_engine *e, int file, int64_t rec_lo, int64_t rec_hi, int32_t tid,
                    int64_t start, int64_t end, uint8_t strand, uint8_t *mask);

/* reads_out for EVERY segment of a plan in one pass: which reads the reference's map function appends for each
 * segment (genome_array.py:800-823 fetch -> strand -> filters -> map_fn; `get_reads` :834-859; bin/psite.py:182 and
 * bin/phase_by_size.py:187 ask per region).  A CSR over (segment, file) pairs, pair index = segment * n_files + file:
 * offsets[n_segments * n_files + 1]; the reads of a pair are listed in file order, so a segment's reads come out in
 * the reference's fetch order (file-major).  pc_mapped_reads_batch computes the lists on the device and returns the
 * offsets and *total = offsets[last]; pc_read_mapped_reads copies the record indices (within their file) out. */
int pc_mapped_reads_batch(pc_engine *e, pc_plan *p, int64_t *offsets, int64_t *total);
int pc_read_mapped_reads(pc_engine *e, pc_plan *p, uint32_t *rec, int64_t total);

/* HIP-event timing of pc_count on the engine's stream.  level 0 (default): no events are
 * recorded; 1: the whole call and the histogram / center kernel; 2: every phase.  Each
 * recorded event costs a few microseconds of stream time, which is why it is opt-in. */
int pc_set_profiling(pc_engine *e, int level);
/* last timed pc_count, milliseconds: [0] total, [1] work-list, [2] histogram/center kernel,
 * [3] long reads, [4] gather, [5] zero-fill ([1],[3],[4],[5] are 0 below level 2).
 * Returns number of entries written. */
int pc_last_timing(pc_engine *e, double *ms, int n);
/* algorithmic bytes of the last pc_count (SURVEY.md section 8d formula) */
int64_t pc_last_algorithmic_bytes(pc_engine *e);
/* Measurement helper for the center rule (CenterMapFactory.__call__, map_factories.pyx:200-265): counts `plan` once
 * more in a diagnostic launch and reports how many replay steps (one step = entry j of each 16-lane row applied to the
 * row: 3 single-rate vector instructions + one v_fmac_f64) and how many waves the launch executed -- the numerator
 * of the kernel's vector-issue bound that bench.py reports beside the HBM fraction.  No reference counterpart. */
int pc_center_replay_steps(pc_engine *e, pc_plan *p, int64_t *steps, int64_t *waves);
/* Row fill of the center kernel's dispatch list (plans over one alignment file, after a center-rule count): the four
 * 16-lane rows of a wave replay in lock-step, one step per entry of the LONGEST row, so a step does useful work for
 * row_entries / row_slots of its four rows -- row_entries = entries of all rows of all dispatch entries, row_slots =
 * 4 x replay steps.  A measurement helper like pc_center_replay_steps; no reference counterpart. */
int pc_center_row_fill(pc_engine *e, pc_plan *p, int64_t *row_entries, int64_t *row_slots);
/* Measured streaming rates of this GPU (GB/s) for the access patterns of the tile kernel: 16-byte
 * contiguous loads per lane and 8-byte contiguous stores per lane over a buffer of `bytes` bytes
 * (>= 1 MiB; use several hundred MB to get past the 256 MiB Infinity Cache).  The second roofline
 * denominator of SURVEY.md 8(d); no reference counterpart. */
int pc_stream_probe(pc_engine *e, int64_t bytes, int iters, double *read_gbps, double *write_gbps);

/* ---- compressed BAM on the GPU (extends SURVEY.md 8(f1)).
 * Replaces, for a whole coordinate-sorted file, what the reference gets from pysam / htslib: `pysam.AlignmentFile(X, "rb")`
 * + `fetch` + `read.positions` / `read.is_reverse` / `.mapped` (plastid/genomics/genome_array.py:660, 669, 690, 800-815),
 * i.e. htslib's BGZF reader (kent/src/htslib/bgzf.c:292-340 block header, :421-530 inflate + CRC check), bam_read1
 * (kent/src/htslib/sam.c) and the CIGAR operation table (kent/src/htslib/htslib/sam.h:64-104).
 * `image` = the bytes of the .bam file (host memory; e.g. an mmap).  The BGZF members are inflated on the GPU (one wave
 * per member), the BAM records found and decoded there; what comes back are the packed columns pc_add_alignment_file
 * takes -- bit-identical to those of the host decoder (plastid_amd/csrc/bam_stager.cpp), same checks, same messages.
 * Errors: PC_ERR_ARG (not a BGZF / BAM file, damaged member, CRC mismatch, corrupt record), PC_ERR_UNSORTED (not
 * coordinate sorted -- pysam raises ValueError at fetch, genome_array.py:784-787). */
typedef struct pc_bam pc_bam;
int pc_bam_open(pc_engine *e, const void *image, int64_t size, const char *name /* for messages; may be NULL */, pc_bam **out);
/* counts[0] staged (placed) records, [1] runs of the multi-run ones, [2] mapped (flag 0x4 unset: pysam's .mapped),
 * [3] all records, [4] wide records, [5] BGZF members, [6] inflated bytes, [7] members whose first-record guess was redone */
int pc_bam_counts(pc_bam *b, int64_t *counts8);
/* milliseconds on the engine's stream: [0] upload of the image, [1] inflate + CRC, [2] record chain, [3] fields + columns */
int pc_bam_timing(pc_bam *b, double *ms4);
int pc_bam_nref(pc_bam *b);
const char *pc_bam_ref_name(pc_bam *b, int i);
int32_t pc_bam_ref_length(pc_bam *b, int i);
/* the columns, to caller-owned arrays of counts[0] / counts[1] / counts[4] elements (see pc_add_alignment_file_wide) */
int pc_bam_read(pc_bam *b, int32_t *tid, int32_t *pos, uint16_t *alen, uint8_t *flags, uint8_t *nblk, int32_t *blk_start,
                int32_t *blk_len, int64_t *wide_idx, int32_t *wide_alen, int32_t *wide_nblk);
/* the SAM FLAG word, MAPQ and l_seq of the same records (pysam: read.flag, .mapping_quality, .query_length -- what a
 * filter function may look at, genome_array.py:697-722); any pointer may be NULL */
int pc_bam_read_sam(pc_bam *b, uint16_t *flag, uint8_t *mapq, int32_t *lseq);
/* ... and their NH:i tags (0: none), what read.get_tag("NH") / read.has_tag("NH") answer from (pysam AlignedSegment; the
 * walk over the auxiliary fields is htslib's bam_aux_get, kent/src/htslib/sam.c). */
int pc_bam_read_nh(pc_bam *b, uint16_t *nh);
int pc_bam_close(pc_bam *b);
/* decode `image` and stage it as one more alignment file of the engine (reference ids = the file's own reference
 * list, which must be that of the files staged before it); *mapped (optional) = the file's mapped-read count */
int pc_add_alignment_bam(pc_engine *e, const void *image, int64_t size, const char *name, int64_t *mapped);
/* The same two calls for a file named by path (what `pysam.AlignmentFile(path, "rb")` takes, genome_array.py:660): the
 * library maps the file itself -- the pages are touched by all host threads at once instead of one soft fault after the
 * other, and the mapping is taken down on a thread of its own after the call has returned. */
int pc_bam_open_path(pc_engine *e, const char *path, pc_bam **out);
int pc_add_alignment_bam_path(pc_engine *e, const char *path, int64_t *mapped);
/* REGION reads: what `AlignmentFile.fetch(reference, start, end)` returns (genome_array.py:800-809; htslib's iterator,
 * kent/src/htslib/hts.c:1924-1960) for a set of regions, decoded on the GPU.  The caller resolves the regions through the
 * file's BAI index (bins + 16 kb linear index: plastid_amd/csrc/bam_stager.cpp pb_resolve_regions) into
 *   voff_begin, voff_end   the span of virtual offsets (file offset of a BGZF member << 16 | offset in its payload) that
 *                          holds every chunk of every region; 0, 0: nothing overlaps (the header alone is read);
 *   nreg, tid, beg, end    the regions themselves by reference id, ascending by (tid, beg), merged (none overlap).
 * Only the members of the span -- and the leading ones that hold the header -- are uploaded and inflated; the record
 * chain starts at the record the index points to; a placed record stays iff pos < end && endpos > beg for one of the
 * regions (htslib's rule, as the host reader applies it).  counts[2] / *mapped = mapped records among those kept (the
 * whole file's count lives in the index).  One rank of a multi-GPU job stages its genome range of ONE shared file this
 * way.  Errors: as pc_bam_open, plus PC_ERR_ARG when the span does not fit the file (a foreign index). */
int pc_bam_open_span(pc_engine *e, const char *path, uint64_t voff_begin, uint64_t voff_end, int nreg, const int32_t *tid,
                     const int64_t *beg, const int64_t *end, pc_bam **out);
int pc_add_alignment_bam_span(pc_engine *e, const char *path, uint64_t voff_begin, uint64_t voff_end, int nreg, const int32_t *tid,
                              const int64_t *beg, const int64_t *end, int64_t *mapped);

#ifdef __cplusplus
}
#endif
#endif /* PLASTID_COUNTS_H */
