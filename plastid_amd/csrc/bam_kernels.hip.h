// bam_kernels.hip.h -- compressed BAM on the GPU (round 4; extends SURVEY 8(f1)).
//
// The reference reaches its reads through pysam / htslib: `AlignmentFile.fetch` over BGZF
// (plastid/genomics/genome_array.py:800-809).  A BAM file is a series of BGZF members, each an independent raw DEFLATE
// stream of at most 64 KiB of payload (RFC 1951 / 1952; read by the vendored kent/src/htslib/bgzf.c:292-340, 421-530),
// whose concatenated payloads are the BAM stream: header, then records  {block_size:u32, refID:i32, pos:i32,
// l_read_name:u8, mapq:u8, bin:u16, n_cigar_op:u16, flag:u16, l_seq:i32, next_refID, next_pos, tlen, read_name,
// cigar[n_cigar_op]:u32 (op = v & 15, len = v >> 4; ops MIDNSHP=X, kent/src/htslib/htslib/sam.h:64-104), seq, qual,
// tags}  (kent/src/htslib/sam.c bam_read1).  Host inflate (zlib / libdeflate on the 16 CPUs a GPU box grants) runs at
// ~11 GB/s: 5.4e7 reads/s from a realistic file, against 6e11 at kernel scope.  Here the file image goes to HBM as it
// is and
//   k_bgzf_inflate    ONE WAVE PER MEMBER inflates it: Huffman tables and the last 2 KiB of output in LDS; block headers
//                     and code lengths read wave-uniform, the symbols of a block a batch of 512 bit offsets at a time
//                     (every lane looks the symbol up at 8 offsets, a walk finds the real starts, positions / literals /
//                     matches per symbol in parallel: see "the symbols of a block" below); a match that reaches further
//                     back than the LDS window reads its source from the member's output in HBM.  43 GB/s on records
//                     as an aligner writes them (the wave-uniform symbol decoder it replaced, PC_BGZF_SERIAL=1: ~27);
//   k_bgzf_crc        CRC-32 of every member's payload, 64 slices per member combined by a shift operator;
//   k_bam_chain       one wave per member finds where the first BAM record of the member starts (a guess: the first
//                     offset from which a few records in a row look like records) and walks the chain of length prefixes
//                     to the first record start of the NEXT member, noting every start; the host only confirms that
//                     the guesses chain (and restarts the few members whose guess did not);
//   k_bam_fields      one thread per record: fixed-offset fields, CIGAR -> aligned runs (sam.h:64-104), the checks of
//                     the host decoder (bam_stager.cpp decode_span_cols) as an error code;
//   k_bam_order / k_bam_scan_inputs / k_bam_columns   sort order and unplaced reads, the inputs of the two exclusive sums
//                     (staged index, run offset), and the packed columns with the runs of the multi-run records.
// The columns that come out (tid, pos, alen, flags, nblk, runs) are what pc_add_alignment_file takes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pcbam {

constexpr int kInflWG = 64;                 // one wave per BGZF member
#ifndef PC_BGZF_WINDOW
#define PC_BGZF_WINDOW 2048
#endif
constexpr int kWinBytes = PC_BGZF_WINDOW;   // the part of the DEFLATE window (RFC 1951: distances up to 32 768) kept in LDS; further back: HBM
constexpr int kLitRoot = 9, kDistRoot = 6;  // first-level table bits (zlib's choice: enough.c bounds 852 / 592 entries)
// table capacities: zlib's ENOUGH_LENS / ENOUGH_DISTS (inftrees.h: 286 symbols, root 9, 15 bits -> 852 entries; 30 symbols,
// root 6 -> 592), first level included -- build_table sizes a second-level table by the longest code under its prefix,
// as inflate_table does for a complete code, so no valid stream needs more.  (Round 6: 1024 / 640 before; with the
// symbol chain compacted in place, 16-bit entries and 256 bytes of staged input a wave's LDS is 7.1 KiB instead of 13.)
constexpr int kLitEntries = 852, kDistEntries = 592;
#ifndef PC_BGZF_IN
#define PC_BGZF_IN 256
#endif
constexpr int kInBytes = PC_BGZF_IN;       // compressed input staged in LDS (two halves)
constexpr uint32_t kInHalf = kInBytes / 2;
constexpr int kFlush = kWinBytes / 4;       // the window goes to HBM in pieces of this size

struct Member {
    uint64_t coff;     // offset of the raw DEFLATE stream in the file image (behind the gzip header)
    uint32_t clen;     // its length (the 8-byte trailer excluded)
    uint32_t ulen;     // ISIZE: bytes it inflates to
    uint64_t uoff;     // where they go in the inflated stream
    uint32_t crc;      // CRC-32 of the payload (gzip trailer)
    uint32_t pad;
};

// error codes of k_bgzf_inflate (per member)
enum { kInfOk = 0, kInfBadBlockType = 1, kInfBadStored = 2, kInfBadCodeLengths = 3, kInfOverSubscribed = 4, kInfBadSymbol = 5,
       kInfBadDistance = 6, kInfOverrun = 7, kInfShort = 8, kInfInputOverrun = 9, kInfCrc = 10 };

// ---- table entries (uint16, round 6: half the LDS of the uint32 ones).  Bits 0-3: code bits to consume (0: no such
// code).  What a lookup needs next sits in the entry ready to use, so that the per-offset lookup of a batch does no
// arithmetic on symbols (RFC 1951 3.2.5: length = 3 + v + extra bits, distance = 1 + (m << e) + e extra bits):
//   literal / length table   bit 15: a length -- bits 4-11 v = base - 3, bits 12-14 its number of extra bits
//                            else bit 14: pointer to a second-level table -- bits 4-13 its first entry, bits 0-3 its index bits
//                            else bit 13: end of block (bit 4 = 0) or a symbol that is never valid (bit 4 = 1)
//                            else a literal -- bits 4-11 the byte (12-14 zero: "no extra bits")
//   distance table           bit 14: pointer, as above;  bit 13: a symbol that is never valid (30, 31);
//                            else bits 4-7 e, bits 8-9 m  (symbols 0-3: m = symbol, e = 0; others m = 2 + (s & 1), e = (s >> 1) - 1)
//   code-length table        bits 4-8 the symbol (0-18)
typedef uint16_t tab_t;
constexpr uint32_t kEntLen = 0x8000u, kEntPtr = 0x4000u, kEntSpecial = 0x2000u, kEntBadBit = 0x10u;
__device__ __forceinline__ uint32_t ent_bits(uint32_t e) { return e & 15u; }
__device__ __forceinline__ bool ent_is_ptr(uint32_t e) { return (e & (kEntLen | kEntPtr)) == kEntPtr; }
__device__ __forceinline__ uint32_t ent_ptr_start(uint32_t e) { return (e >> 4) & 1023u; }
__device__ __forceinline__ uint32_t ent_v8(uint32_t e) { return (e >> 4) & 255u; }
__device__ __forceinline__ uint32_t mk_ptr(uint32_t index_bits, uint32_t start) { return index_bits | (start << 4) | kEntPtr; }
// leaf entries without their bit count
__device__ __forceinline__ uint32_t mk_lit_leaf(int s) {
    if (s < 256) return (uint32_t)s << 4;
    if (s == 256) return kEntSpecial;
    if (s > 285) return kEntSpecial | kEntBadBit;                       // 286 / 287: never valid
    const uint32_t i = (uint32_t)(s - 257);
    const uint32_t e = i < 8u ? 0u : (i >> 2) - 1u;
    const uint32_t v = i == 28u ? 255u : (i < 8u ? i : (4u + (i & 3u)) << e);      // base - 3
    return kEntLen | (v << 4) | ((i == 28u ? 0u : e) << 12);
}
__device__ __forceinline__ uint32_t mk_dist_leaf(int s) {
    if (s >= 30) return kEntSpecial | kEntBadBit;
    const uint32_t e = s < 4 ? 0u : ((uint32_t)s >> 1) - 1u, m = s < 4 ? (uint32_t)s : 2u + ((uint32_t)s & 1u);
    return (e << 4) | (m << 8);
}

__device__ const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

__device__ __forceinline__ uint32_t bitrev(uint32_t c, int len) { return __brev(c) >> (32 - len); }

// Canonical Huffman code of `n` symbols with code lengths `lens` (0: unused) -> two-level decode table with `root`
// first-level bits.  Run by the whole wave (wave-uniform control flow; the fills are spread over the lanes).
// KIND 0: literal/length alphabet (symbols 0-255 literals, 256 end of block, 257-285 lengths); 1: distances; 2: the
// code-length alphabet (every symbol a "literal").  Returns 0, or an error code (over-subscribed or incomplete code --
// a single one-bit code is allowed, as in zlib -- or a table that does not fit `cap` entries).
// Small arrays live in LDS (`ws`): run-time indices into private arrays would put them into scratch memory.
struct TableScratch {
    int count[16];
    int next[16];
    uint16_t code[320];
    uint8_t subbits[512];
};

template <int KIND>
__device__ int build_table(const uint8_t *lens, int n, int root, tab_t *table, int cap, TableScratch *ws, int lane) {
    // symbols per code length: lane L (1..15) counts the symbols of length L
    if (lane < 16) {
        int c = 0;
        if (lane >= 1) for (int s = 0; s < n; ++s) c += lens[s] == lane ? 1 : 0;
        ws->count[lane] = c;
    }
    __syncthreads();
    int left = 1, maxlen = 0, nsym = 0, code = 0;
    for (int len = 1; len <= 15; ++len) {          // (uniform: LDS broadcasts)
        const int c = ws->count[len];
        left = (left << 1) - c;
        if (left < 0) return kInfOverSubscribed;
        if (c) maxlen = len;
        nsym += c;
        code = (code + ws->count[len - 1]) << 1;   // first code of this length
        if (lane == 0) ws->next[len] = code;
    }
    // no symbols at all (zlib inftrees.c: `if (max == 0)` -> a table of invalid-code markers, no error): a literal-only
    // dynamic block may declare one distance code of length zero; any lookup in the cleared table is a "bad symbol"
    if (maxlen != 0 && left > 0 && (KIND == 2 || maxlen != 1)) return kInfOverSubscribed;   // incomplete code: only a single one-bit code may be (zlib inftrees.c: `left > 0 && (type == CODES || max != 1)`)
    // clear the first level (an incomplete distance code leaves holes: they decode as "bad symbol")
    for (int i = lane; i < (1 << root); i += 64) { table[i] = (tab_t)0; if (i < 512) ws->subbits[i] = 0; }
    __syncthreads();
    if (nsym == 0) return kInfOk;
    // codes in symbol order within a length (lane 0, serial: a few hundred symbols per block); the longest code of
    // every first-level prefix sizes its second-level table
    if (lane == 0) {
        for (int s = 0; s < n; ++s) {
            const int len = lens[s];
            if (!len) continue;
            const int c = ws->next[len];
            ws->next[len] = c + 1;
            ws->code[s] = (uint16_t)c;
            if (len > root) {
                const uint32_t pre = bitrev((uint32_t)c, len) & ((1u << root) - 1u);
                if (ws->subbits[pre] < len - root) ws->subbits[pre] = (uint8_t)(len - root);
            }
        }
    }
    __syncthreads();
    if (maxlen > root) {   // allocate the second-level tables (serial over the prefixes; uniform)
        int used = 1 << root;
        for (int pre = 0; pre < (1 << root); ++pre) {
            const int sb = ws->subbits[pre];
            if (!sb) continue;
            if (used + (1 << sb) > cap) return kInfBadCodeLengths;
            if (lane == 0) table[pre] = (tab_t)mk_ptr((uint32_t)sb, (uint32_t)used);
            for (int i = lane; i < (1 << sb); i += 64) table[used + i] = (tab_t)0;
            used += 1 << sb;
        }
        __syncthreads();
    }
    for (int s = 0; s < n; ++s) {
        const int len = lens[s];
        if (!len) continue;
        const uint32_t rc = bitrev((uint32_t)ws->code[s], len);
        uint32_t e;
        e = KIND == 0 ? mk_lit_leaf(s) : (KIND == 1 ? mk_dist_leaf(s) : (uint32_t)s << 4);
        if (len <= root) {
            e |= (uint32_t)len;
            for (uint32_t i = rc + ((uint32_t)lane << len); i < (1u << root); i += 64u << len) table[i] = (tab_t)e;
        } else {
            const uint32_t pre = rc & ((1u << root) - 1u);
            const uint32_t pe = table[pre];
            const uint32_t base = ent_ptr_start(pe), sb = ent_bits(pe);
            const int sl = len - root;
            e |= (uint32_t)sl;
            for (uint32_t i = (rc >> root) + ((uint32_t)lane << sl); i < (1u << sb); i += 64u << sl) table[base + i] = (tab_t)e;
        }
    }
    __syncthreads();
    return kInfOk;
}

// CRC-32 (RFC 1952) of `n` window bytes starting at window offset `at`: 64 lanes x contiguous slices with a 256-entry
// table in LDS, combined in lane order.  (What htslib checks per member, bgzf.c:421-530.)
__device__ __forceinline__ uint32_t crc_byte(const uint32_t *tab, uint32_t crc, uint32_t b) { return tab[(crc ^ b) & 0xffu] ^ (crc >> 8); }

struct InflateShared {
    tab_t lit[kLitEntries];
    tab_t dist[kDistEntries];
    uint32_t in[kInBytes / 4];
    uint8_t win[kWinBytes];
};

// ---- the symbols of a block, a BATCH of bit offsets at a time (k_bgzf_inflate<true>)
// A wave-uniform decode spends ~45 scalar instructions per symbol, and the CU's one scalar unit is shared by every
// resident wave: more waves per CU stopped paying at ~30 GB/s.  What a symbol at bit offset `i` is -- and where the
// next one starts -- depends only on the bits from `i` on and on the block's tables, so all of it can be looked up
// for EVERY bit offset of the next kBatchBits at once, one lane per 8 consecutive offsets, without a branch:
// sym[i] = {bits consumed, kind, literal byte | match length and distance}.  Which offsets are real symbol starts is
// then a walk over sym[] from offset 0 (one LDS read and a dozen instructions per symbol, nothing else on the
// dependent chain), and everything after it is per SYMBOL, one lane each: output positions by a prefix sum,
// literals stored in parallel, matches whose source lies before the chunk copied in parallel (one per lane), the few
// whose source is inside the chunk in stream order, each by the whole wave.
// (Measured on the way: reachability by pointer doubling over all 512 offsets instead of the walk, with the
// post-processing per offset -- correct, and at ~3 700 vector instructions per batch only 13 % faster than the uniform
// decode.)
// Experiment hook (never set in the product build): -DPC_BGZF_SKIP=<mask> leaves out 1 the copies of matches whose source
// lies before the chunk, 2 the matches that read what their own chunk writes, 4 the literal stores, 8 the flushes to HBM,
// 16 the symbol lookup of the batch (every offset reads as an 8-bit literal), 32 the whole kernel (the lap is then the
// upload) -- the output is then wrong (the CRC check says so), the
// control flow is not: it depends on the compressed stream alone.  PC_BAM_TIMING=1 prints the inflate lap before the check.
#ifndef PC_BGZF_SKIP
#define PC_BGZF_SKIP 0
#endif
constexpr int kBatchBits = 512;             // bit offsets per batch: 64 lanes x 8
// sym: bits 0-7 FOUR TIMES the bits consumed (the walk adds the byte to its LDS address as it is), 8-15 byte | length - 3,
// 16-30 distance - 1, bit 31 a match; no match: bit 30 end of block or no symbol at all (then bit 29), else a literal
constexpr uint32_t kSymMatch = 1u << 31, kSymSpecial = 1u << 30, kSymBadBit = 1u << 29;
__device__ __forceinline__ bool sym_is_match(uint32_t sv) { return (sv >> 31) != 0u; }
__device__ __forceinline__ bool sym_is_special(uint32_t sv) { return (sv >> 30) == 1u; }
__device__ __forceinline__ bool sym_is_lit(uint32_t sv) { return (sv >> 30) == 0u; }
__device__ __forceinline__ uint32_t sym_bits(uint32_t sv) { return (sv & 0xffu) >> 2; }

struct BatchShared {
    // what starts at bit offset i, were it a symbol start -- and, once the walk has passed, the symbols that ARE real, in
    // stream order, compacted IN PLACE: the k-th real symbol starts at an offset >= k (a symbol is at least one bit), so
    // the walk writes entry k behind its own read position and never over an offset it has yet to read
    uint32_t sym[kBatchBits];
};

struct HeaderShared {
    uint8_t lens[352];
    TableScratch ws;
};

// What the symbols at the lane's 8 bit offsets are, by the block's tables ({A, B, C}: the 96 bits from the first of
// them on).  Straight-line -- the lanes of a wave look at different offsets, so every path would be walked anyway;
// second-level lookups that do not apply read entry 0 -- and in stages, each stage's 8 table reads in flight
// together (the empty asm statements pin the stage boundaries: left alone, the compiler sinks the distance lookups
// into a branch and waits for every read on its own).
#define PC_PIN8(x) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]))
#define PC_PING(x) do { if constexpr (G == 8) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4 % G]), "+v"(x[5 % G]), "+v"(x[6 % G]), "+v"(x[7 % G])); \
                        else asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])); } while (0)
template <int G, int T0>
__device__ __forceinline__ void symbols_at(const tab_t *lit, const tab_t *dist, uint32_t A, uint32_t B, uint32_t C, uint32_t (&out)[8]) {
    uint32_t lo[G], w2[G], e[G], x[G], tot2[G], val[G];
#pragma unroll
    for (int t = 0; t < G; ++t) {
        lo[t] = (uint32_t)((((unsigned long long)B << 32) | A) >> (T0 + t));
        e[t] = lit[lo[t] & ((1u << kLitRoot) - 1u)];
    }
    PC_PING(e);
#pragma unroll
    for (int t = 0; t < G; ++t)
        x[t] = lit[ent_is_ptr(e[t]) ? ent_ptr_start(e[t]) + __builtin_amdgcn_ubfe(lo[t], (uint32_t)kLitRoot, ent_bits(e[t])) : 0u];
    PC_PING(x);
#pragma unroll
    for (int t = 0; t < G; ++t) {
        const bool two = ent_is_ptr(e[t]);
        e[t] = two ? x[t] : e[t];
        const uint32_t tot = (two ? (uint32_t)kLitRoot : 0u) + ent_bits(e[t]);                       // <= 15
        const uint32_t xb = (e[t] & kEntLen) ? (e[t] >> 12) & 7u : 0u;                               // (literals: 0)
        val[t] = ent_v8(e[t]) + __builtin_amdgcn_ubfe(lo[t], tot, xb);                               // the byte | the match length - 3
        tot2[t] = tot + xb;                                                                          // <= 20
        const uint32_t hi = (uint32_t)((((unsigned long long)C << 32) | B) >> (T0 + t));
        w2[t] = (uint32_t)((((unsigned long long)hi << 32) | lo[t]) >> tot2[t]);                     // (one v_alignbit: tot2 < 32)
        x[t] = dist[w2[t] & ((1u << kDistRoot) - 1u)];
    }
    PC_PING(x);
    uint32_t y[G];
#pragma unroll
    for (int t = 0; t < G; ++t)
        y[t] = dist[(x[t] & kEntPtr) ? ent_ptr_start(x[t]) + __builtin_amdgcn_ubfe(w2[t], (uint32_t)kDistRoot, ent_bits(x[t])) : 0u];
    PC_PING(y);
#pragma unroll
    for (int t = 0; t < G; ++t) {
        const bool dtwo = (x[t] & kEntPtr) != 0u;
        const uint32_t d = dtwo ? y[t] : x[t];
        const uint32_t dt = (dtwo ? (uint32_t)kDistRoot : 0u) + ent_bits(d);                         // <= 15
        const uint32_t dxb = (d >> 4) & 15u;                                                         // <= 13
        const uint32_t dd1 = (((d >> 8) & 3u) << dxb) + __builtin_amdgcn_ubfe(w2[t], dt, dxb);       // the distance - 1
        const bool dok = ent_bits(d) != 0u && (d & (kEntPtr | kEntSpecial)) == 0u;
        const uint32_t bad = (1u << 2) | kSymSpecial | kSymBadBit;
        uint32_t r = (tot2[t] << 2) | (val[t] << 8);                                                 // a literal (xb = 0: tot2 = tot)
        r = (e[t] & kEntLen) ? (dok ? ((tot2[t] + dt + dxb) << 2) | kSymMatch | (val[t] << 8) | (dd1 << 16) : bad) : r;
        r = (e[t] & (kEntLen | kEntSpecial)) == kEntSpecial ? ((e[t] & kEntBadBit) ? bad : (tot2[t] << 2) | kSymSpecial) : r;
        out[T0 + t] = ent_bits(e[t]) == 0u ? bad : r;
    }
}

#ifndef PC_BGZF_GROUP
#define PC_BGZF_GROUP 4
#endif
__device__ __forceinline__ void symbols_at8(const tab_t *lit, const tab_t *dist, uint32_t A, uint32_t B, uint32_t C, uint32_t (&out)[8]) {
    if constexpr (PC_BGZF_GROUP == 8) symbols_at<8, 0>(lit, dist, A, B, C, out);
    else { symbols_at<4, 0>(lit, dist, A, B, C, out); symbols_at<4, 4>(lit, dist, A, B, C, out); }
}

// One wave inflates one BGZF member.  Block headers, code lengths and stored blocks are read wave-uniform (every lane
// holds the same bit buffer and positions: no divergence, table reads are LDS broadcasts); the symbols of a block go
// batch-wise through the lanes (BATCH, above) or, BATCH = false, one by one through the same uniform reader (round 4's
// first kernel, kept for comparison: PC_BGZF_SERIAL=1).  (The workgroup IS the wave: __syncthreads() orders the LDS
// traffic of its lanes and costs no cross-wave barrier.)
// Occupancy (round 6): 7.1 KiB of LDS allow 22 waves per CU; the registers are held to 96 (five waves per SIMD) by looking
// the batch's symbols up in two groups of four offsets instead of one of eight (PC_BGZF_GROUP).  20 M aligner-like
// records, 2.39 GB inflated, lap `upload + inflate + crc`: 32-bit entries, four waves per SIMD 25.8 - 26.1 ms; 16-bit
// entries holding the bare symbol (bases and extra-bit counts computed per offset) at four waves 27.2 - 27.8, at five
// 24.0 - 24.2, compiled for six (20 bytes of scratch, LDS admits 5.5) 23.9; 16-bit entries holding v / m / e ready to
// use (above), five waves: 22.9 - 23.1; with the seven-instruction walk 21.8 - 22.2 (108 GB/s).  Where the lap goes now
// (-DPC_BGZF_SKIP builds, profiles/r06/bam/sections_final_kernel.txt): the upload alone 11.5 ms (PCIe, overlapped), every
// offset an 8-bit literal and no matches 13.3, the real lookups and symbol counts + 3.8, the match copies + 4.7.
#ifndef PC_BGZF_WAVES
#define PC_BGZF_WAVES 5
#endif
template <bool BATCH>
__global__ __launch_bounds__(kInflWG) __attribute__((amdgpu_waves_per_eu(PC_BGZF_WAVES, 8))) void k_bgzf_inflate(const uint8_t *__restrict__ image, const Member *__restrict__ members, int first_member, int nmembers,
                                                          uint8_t *__restrict__ out, uint32_t *__restrict__ status) {
    __shared__ __attribute__((aligned(16))) InflateShared sh;
    __shared__ __attribute__((aligned(16))) union { HeaderShared hdr; BatchShared bat; } su;
    uint8_t *const s_lens = su.hdr.lens;
    TableScratch &s_ws = su.hdr.ws;
    const int m = first_member + (int)blockIdx.x;   // (the members of one upload piece: pc_bam_open)
    if (m >= nmembers) return;
    if (PC_BGZF_SKIP & 32) { if (threadIdx.x == 0) status[m] = 0u; return; }   // (experiment: the lap without the kernel = the upload)
    const Member mb = members[m];   // (uniform: scalar loads)
    const int lane = threadIdx.x & 63;
    const uint8_t *src = image + mb.coff;
    uint8_t *dst = out + mb.uoff;
    const uint32_t clen = mb.clen, ulen = mb.ulen;
    int err = kInfOk;

    // ---- input: `in` holds the kInBytes of the stream around the read position (ring); refilled a half at a time by
    // the wave when the reader has crossed into the other half (256 bytes: a batch looks 84 bytes ahead of its first bit,
    // and a half is staged before the reader comes within 8 bytes of the staged end -- both fit 128-byte halves)
    uint32_t in_pos = 0;          // next byte of the stream to pull into the bit buffer (always a multiple of 4)
    uint32_t in_loaded = 0;       // bytes of the stream staged so far (multiple of kInHalf)
    auto stage_half = [&]() {     // stage stream bytes [in_loaded, in_loaded + kInHalf)
        const uint32_t base = in_loaded;
        // dwords assembled byte-wise (the stream starts at an arbitrary byte of the image)
        for (int k = lane; k < (int)kInHalf / 4; k += 64) {
            const uint32_t b = base + 4u * (uint32_t)k;
            uint32_t w = 0;
            if (b + 3u < clen) w = (uint32_t)src[b] | ((uint32_t)src[b + 1] << 8) | ((uint32_t)src[b + 2] << 16) | ((uint32_t)src[b + 3] << 24);
            else {
                if (b < clen) w |= (uint32_t)src[b];
                if (b + 1u < clen) w |= (uint32_t)src[b + 1] << 8;
                if (b + 2u < clen) w |= (uint32_t)src[b + 2] << 16;
            }
            sh.in[((base >> 2) + (uint32_t)k) & (kInBytes / 4 - 1)] = w;
        }
        in_loaded += kInHalf;
        __syncthreads();
    };
    stage_half();
    stage_half();
    unsigned long long bb = 0;    // bit buffer
    int nb = 0;                   // valid bits in it
    auto refill = [&]() {         // at least 32 valid bits afterwards (zeros behind the end of the stream)
        if (nb <= 32) {
            // Stage the next half only when the reader is within kInHalf - 8 bytes of the staged end: staging overwrites the
            // ring slots of [in_loaded - kInBytes, in_loaded - kInHalf), and the 8 bytes below in_pos must stay in the ring --
            // the bit buffer may still hold up to 64 unread bits of them when the batch decoder takes over and re-reads the
            // stream from its bit position p = 8 in_pos - nb (invariant: in_pos - 8 >= in_loaded - kInBytes whenever a
            // block's symbols start).  (Zeros behind the end of the stream.)
            if (in_pos + kInHalf - 8u >= in_loaded) stage_half();
            // (every lane reads the same word: readfirstlane tells the compiler so, and what follows stays on the scalar unit)
            const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.in[(in_pos >> 2) & (kInBytes / 4 - 1)]);
            bb |= (unsigned long long)w << nb;
            nb += 32;
            in_pos += 4u;
        }
    };
    auto take = [&](int n) -> uint32_t {   // n <= 16
        const uint32_t v = (uint32_t)bb & ((1u << n) - 1u);
        bb >>= n;
        nb -= n;
        return v;
    };

    uint32_t pos = 0;             // bytes produced
    uint32_t flushed = 0;         // bytes written to HBM
    auto flush_to = [&](uint32_t upto) {   // window bytes [flushed, upto) -> HBM; both multiples of 16 except at the very end
        __syncthreads();
        if (PC_BGZF_SKIP & 8) { flushed = upto; return; }
        for (uint32_t b = flushed + 16u * (uint32_t)lane; b < upto; b += 16u * 64u) {
            if (b + 16u <= upto && ((mb.uoff + b) & 15u) == 0u) {
                const uint4 v = *(const uint4 *)&sh.win[b & (kWinBytes - 1)];
                *(uint4 *)(dst + b) = v;
            } else {
                for (uint32_t k = b; k < upto && k < b + 16u; ++k) dst[k] = sh.win[k & (kWinBytes - 1)];
            }
        }
        flushed = upto;
    };

    bool last = false;
    while (!last && err == kInfOk) {
        refill();
        last = take(1) != 0u;
        const uint32_t type = take(2);
        if (type == 0u) {
            // stored block: skip to the byte boundary, LEN / NLEN, raw bytes
            take(nb & 7);
            refill();
            const uint32_t len = take(16);
            refill();
            const uint32_t nlen = take(16);
            if ((len ^ 0xffffu) != nlen) { err = kInfBadStored; break; }
            if (pos + len > ulen) { err = kInfOverrun; break; }
            // the bit buffer holds whole bytes now: give them back to the stream position
            uint32_t sp = in_pos - (uint32_t)(nb >> 3);
            bb = 0; nb = 0;
            if (sp + len > clen) { err = kInfInputOverrun; break; }
            // (a stored block can be twice the window: copied and flushed piece by piece)
            for (uint32_t done = 0; done < len;) {
                const uint32_t piece = len - done < (uint32_t)kFlush ? len - done : (uint32_t)kFlush;
                for (uint32_t k = lane; k < piece; k += 64) sh.win[(pos + k) & (kWinBytes - 1)] = src[sp + done + k];
                __syncthreads();
                pos += piece;
                done += piece;
                while (pos - flushed >= (uint32_t)kFlush + 16u) flush_to((flushed + kFlush) & ~15u);
            }
            sp += len;
            // restart the staged input at the new position (dword aligned below it; the odd bytes are dropped from the bit buffer)
            in_pos = sp & ~3u;
            in_loaded = in_pos & ~(kInHalf - 1u);
            stage_half();
            stage_half();
            refill();
            take((int)((sp & 3u) * 8u));
            continue;
        }
        if (type == 3u) { err = kInfBadBlockType; break; }
        if (type == 1u) {
            // fixed Huffman codes (RFC 1951 3.2.6)
            for (int s = lane; s < 288; s += 64) s_lens[s] = s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8));
            __syncthreads();
            err = build_table<0>(s_lens, 288, kLitRoot, sh.lit, kLitEntries, &s_ws, lane);
            if (err) break;
            for (int s = lane; s < 32; s += 64) s_lens[s] = 5;
            __syncthreads();
            err = build_table<1>(s_lens, 32, kDistRoot, sh.dist, kDistEntries, &s_ws, lane);   // (32 five-bit codes: 30 and 31 never valid)
            if (err) break;
        } else {
            // dynamic codes: HLIT, HDIST, HCLEN, the code-length code, then the two length lists (RFC 1951 3.2.7)
            refill();
            const int hlit = (int)take(5) + 257, hdist = (int)take(5) + 1, hclen = (int)take(4) + 4;
            if (hlit > 286 || hdist > 30) { err = kInfBadCodeLengths; break; }
            if (lane < 19) s_lens[lane] = 0;
            __syncthreads();
            for (int i = 0; i < hclen; ++i) {
                refill();
                const uint32_t v = take(3);
                if (lane == 0) s_lens[kClOrder[i]] = (uint8_t)v;
            }
            __syncthreads();
            // the code-length code (19 symbols, lengths <= 7) decodes through a 7-bit table in the distance table's space
            err = build_table<2>(s_lens, 19, 7, sh.dist, kDistEntries, &s_ws, lane);
            if (err) break;
            int got = 0, prev = 0;
            const int want = hlit + hdist;
            while (got < want && err == kInfOk) {
                refill();
                const uint32_t ce = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.dist[(uint32_t)bb & 127u]);
                const int len = (int)ent_bits(ce), sym = (int)(ce >> 4);
                if (len == 0) { err = kInfBadCodeLengths; break; }
                take(len);
                int rep = 1, val = sym;
                if (sym == 16) { if (got == 0) { err = kInfBadCodeLengths; break; } rep = 3 + (int)take(2); val = prev; }
                else if (sym == 17) { rep = 3 + (int)take(3); val = 0; }
                else if (sym == 18) { rep = 11 + (int)take(7); val = 0; }
                if (got + rep > want) { err = kInfBadCodeLengths; break; }
                if (lane == 0) for (int k = 0; k < rep; ++k) s_lens[32 + got + k] = (uint8_t)val;   // (kept clear of the 19 code-length lengths)
                got += rep;
                prev = val;
            }
            if (err) break;
            __syncthreads();
            if (s_lens[32 + 256] == 0) { err = kInfBadCodeLengths; break; }     // no end-of-block code
            err = build_table<0>(s_lens + 32, hlit, kLitRoot, sh.lit, kLitEntries, &s_ws, lane);
            if (err) break;
            // the distance lengths follow the literal/length ones: move them to a 4-byte aligned place of their own
            uint8_t dl = (lane < hdist) ? s_lens[32 + hlit + lane] : 0;
            __syncthreads();
            if (lane < 32) s_lens[lane] = dl;
            __syncthreads();
            err = build_table<1>(s_lens, hdist, kDistRoot, sh.dist, kDistEntries, &s_ws, lane);
            if (err) break;
        }
        // ---- the symbols of the block
        if constexpr (BATCH) {
            BatchShared &bs = su.bat;
            uint32_t p = in_pos * 8u - (uint32_t)nb;       // bit position of the next symbol in the stream
            bool eob = false;
            while (!eob) {
                if (pos - flushed >= (uint32_t)kFlush) flush_to(pos & ~15u);
                // what the batch may produce: everything unflushed plus a maximal match stays inside the LDS window, so that
                // a match source is either whole in the window or whole in what has been flushed
                const uint32_t cap = flushed + (uint32_t)kWinBytes - 258u;
                const uint32_t limit = cap < ulen ? cap : ulen;
                if ((p >> 3) > clen + 8u) { err = kInfInputOverrun; break; }
                while (in_loaded < (p >> 3) + 84u) stage_half();
                __syncthreads();
                // ---- every bit offset's symbol: lane l looks at offsets 8 l .. 8 l + 7
                {
                    const uint32_t a = p + 8u * (uint32_t)lane;
                    const uint32_t di = a >> 5, s0 = a & 31u;
                    const uint32_t d0 = sh.in[di & (kInBytes / 4 - 1)], d1 = sh.in[(di + 1u) & (kInBytes / 4 - 1)];
                    const uint32_t d2 = sh.in[(di + 2u) & (kInBytes / 4 - 1)], d3 = sh.in[(di + 3u) & (kInBytes / 4 - 1)];
                    // the 96 bits from the lane's first offset on
                    const uint32_t A = (uint32_t)((((unsigned long long)d1 << 32) | d0) >> s0), B = (uint32_t)((((unsigned long long)d2 << 32) | d1) >> s0);
                    const uint32_t C = (uint32_t)((((unsigned long long)d3 << 32) | d2) >> s0);
                    uint32_t sy[8];
                    if (PC_BGZF_SKIP & 16) { for (int t = 0; t < 8; ++t) sy[t] = (8u << 2) | ((A >> t) & 0xff00u); }   // (experiment: no lookup -- every offset an 8-bit literal)
                    else symbols_at8(sh.lit, sh.dist, A, B, C, sy);
                    *(uint4 *)&bs.sym[8 * lane] = make_uint4(sy[0], sy[1], sy[2], sy[3]);
                    *(uint4 *)&bs.sym[8 * lane + 4] = make_uint4(sy[4], sy[5], sy[6], sy[7]);
                }
                __syncthreads();
                // ---- which offsets are symbol starts: the walk from offset 0.  It does not look at what it passes: behind an
                // end-of-block code or a symbol that is none it walks on through entries read with the wrong tables (every
                // entry consumes at least a bit, so it ends), and the stage below stops at the first such symbol.
                uint32_t walk, nsym;
                {
                    // seven instructions per symbol (twelve with a scalar bit count and a test for the stop symbols), one LDS
                    // read on the dependent chain; every lane stores the same value to the same place (no mask to set up)
                    uint32_t va = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)bs.sym;
                    const uint32_t vn0 = va, vend = va + 4u * (uint32_t)kBatchBits;   // (the chain is written over the offset table: see BatchShared)
                    uint32_t vn = vn0, vs;
                    asm volatile("1:\n\t"
                                 "ds_read_b32 %[vs], %[va]\n\t"
                                 "s_waitcnt lgkmcnt(0)\n\t"
                                 "ds_write_b32 %[vn], %[vs]\n\t"
                                 "v_add_u32 %[vn], 4, %[vn]\n\t"
                                 "v_add_u32_sdwa %[va], %[va], %[vs] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
                                 "v_cmp_gt_u32 vcc, %[vend], %[va]\n\t"
                                 "s_cbranch_vccnz 1b\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : [va] "+v"(va), [vn] "+v"(vn), [vs] "=&v"(vs)
                                 : [vend] "v"(vend)
                                 : "vcc", "memory");
                    nsym = (uint32_t)__builtin_amdgcn_readfirstlane((int)(vn - vn0)) >> 2;
                    walk = (uint32_t)__builtin_amdgcn_readfirstlane((int)(va - vn0)) >> 2;
                }
                __syncthreads();
                // ---- per symbol, 64 at a time
                uint32_t done = 0;            // bytes produced by the chunks before
                uint32_t p_next = walk;       // where the next batch starts (bits from p), unless a symbol stops this one
                bool stopped = false;
                uint32_t bits = 0;            // bits consumed by the chunks before
                for (uint32_t c0 = 0; c0 < nsym && !stopped; c0 += 64) {
                    const uint32_t idx = c0 + (uint32_t)lane;
                    const bool have = idx < nsym;
                    const uint32_t sv = have ? bs.sym[idx] : 0u;
                    const uint32_t ol = !have ? 0u : (sym_is_lit(sv) ? 1u : (sym_is_match(sv) ? ((sv >> 8) & 255u) + 3u : 0u));
                    // one prefix sum for both: bytes produced (bits 0-15; <= 64 x 258) and bits consumed (16-31; <= 64 x 48)
                    const uint32_t mine = ol | ((sv & 0xfcu) << 14);   // (sv = 0 without a symbol)
                    uint32_t incl2 = mine;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const uint32_t y = (uint32_t)__shfl_up((int)incl2, d, 64);
                        if (lane >= d) incl2 += y;
                    }
                    const uint32_t incl = incl2 & 0xffffu;
                    const uint32_t off = bits + ((incl2 - mine) >> 16);
                    const uint32_t chunk_start = pos + done;
                    const uint32_t q = chunk_start + incl - ol, end = chunk_start + incl;
                    const unsigned long long stopm = __ballot(have && (sym_is_special(sv) || end > limit));
                    const int first = stopm ? __builtin_ctzll(stopm) : 64;
                    const uint32_t chunk_bytes = first < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)(incl - ol), first) : (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    const uint32_t chunk_end = chunk_start + chunk_bytes;
                    const bool emit = have && lane < first;
                    if (!(PC_BGZF_SKIP & 4) && emit && sym_is_lit(sv)) sh.win[q & (kWinBytes - 1)] = (uint8_t)(sv >> 8);
                    // ---- matches: those whose source lies before the chunk are independent of one another (one per lane);
                    // those that read what this chunk writes follow in stream order, each copied by the whole wave
                    const bool ism = emit && sym_is_match(sv);
                    const uint32_t len = ol, dd = ((sv >> 16) & 0x7fffu) + 1u;
                    if (__ballot(ism && dd > q) != 0ull) { err = kInfBadDistance; break; }
                    const uint32_t src = q - dd;
                    const bool dep = ism && src + len > chunk_start;
                    const bool far = ism && chunk_end - src > (uint32_t)kWinBytes;     // (whole in what has been flushed: see `cap`)
                    if (__ballot(far) != 0ull) __builtin_amdgcn_s_waitcnt(0);         // the flush stores have landed
                    __syncthreads();
                    // The independent matches of the chunk, by OUTPUT BYTE: a lane per byte, 64 bytes a round.  (One lane per
                    // match, copying its own bytes, ran every lane for as long as the longest match of the chunk lasted -- with
                    // read names, quality runs and tags that is often a hundred bytes and more: 16 of the kernel's 42 ms on records as
                    // an aligner writes them, scripts/exp_bam_sections.py.)  Which symbol a byte belongs to: the symbols' first
                    // bytes are marked in 64 flag bytes (the offset table of the batch is free by now), and a byte's owner is the
                    // number of marks up to it -- a ballot and a population count; the owner's source and start come by lane permute.
                    if (!(PC_BGZF_SKIP & 1) && __ballot(ism && !dep) != 0ull) {
                        uint8_t *marks = (uint8_t *)bs.sym;   // (entries 0 .. 15 of the chain: this chunk's symbols are in registers, later chunks read from entry 64 on)
                        const uint32_t startrel = emit ? q - chunk_start : 0xffffu;
                        const uint32_t info = (startrel & 0xffffu) | ((ism && !dep) ? 1u << 16 : 0u) | (far ? 1u << 17 : 0u);
                        for (uint32_t base = 0; base < chunk_bytes; base += 64u) {
                            marks[lane] = 0;
                            __syncthreads();
                            if (emit && ol != 0u && startrel >= base && startrel < base + 64u) marks[startrel - base] = 1;
                            __syncthreads();
                            const unsigned long long mk = __ballot(marks[lane] != 0);
                            const uint32_t before = (uint32_t)__popcll(__ballot(emit && ol != 0u && startrel < base));
                            const uint32_t owner = before + (uint32_t)__popcll(mk & (~0ull >> (63 - lane))) - 1u;   // (the chunk's first symbol starts at its byte 0)
                            const uint32_t o_src = (uint32_t)__shfl((int)src, (int)(owner & 63u), 64);
                            const uint32_t o_info = (uint32_t)__shfl((int)info, (int)(owner & 63u), 64);
                            const uint32_t x = base + (uint32_t)lane;
                            if (x < chunk_bytes && ((o_info >> 16) & 1u)) {
                                const uint32_t from = o_src + (x - (o_info & 0xffffu));
                                const uint8_t b = ((o_info >> 17) & 1u) ? __hip_atomic_load(dst + from, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)   // (past this CU's L1, which may hold an older state of the line)
                                                                        : sh.win[from & (kWinBytes - 1)];
                                sh.win[(chunk_start + x) & (kWinBytes - 1)] = b;
                            }
                            __syncthreads();
                        }
                    }
                    __syncthreads();
                    unsigned long long dm = (PC_BGZF_SKIP & 2) ? 0ull : __ballot(dep);
                    while (dm) {
                        const int l = __builtin_ctzll(dm);
                        dm &= dm - 1ull;
                        const uint32_t q1 = (uint32_t)__builtin_amdgcn_readlane((int)q, l), len1 = (uint32_t)__builtin_amdgcn_readlane((int)len, l);
                        const uint32_t dd1 = (uint32_t)__builtin_amdgcn_readlane((int)dd, l);
                        // byte k comes from `dd1` back; where the match overlaps itself the pattern repeats
                        for (uint32_t k = (uint32_t)lane; k < len1; k += 64u) {
                            const uint32_t from = dd1 >= len1 ? q1 - dd1 + k : q1 - dd1 + (k % dd1);
                            sh.win[(q1 + k) & (kWinBytes - 1)] = sh.win[from & (kWinBytes - 1)];
                        }
                        __syncthreads();
                    }
                    done += chunk_bytes;
                    bits += (uint32_t)__builtin_amdgcn_readlane((int)incl2, 63) >> 16;
                    if (first < 64) {         // the symbol that ends the batch: end of block, none at all, or one that does not fit
                        const uint32_t o1 = (uint32_t)__builtin_amdgcn_readlane((int)off, first);
                        const uint32_t e1 = (uint32_t)__builtin_amdgcn_readlane((int)end, first), s1 = (uint32_t)__builtin_amdgcn_readlane((int)sv, first);
                        stopped = true;
                        if (sym_is_special(s1) && (s1 & kSymBadBit)) err = kInfBadSymbol;
                        else if (sym_is_special(s1)) { eob = true; p_next = o1 + sym_bits(s1); }
                        else if (e1 > ulen) err = kInfOverrun;
                        else p_next = o1;
                    }
                }
                if (err) break;
                pos += done;
                p += p_next;
            }
            if (err) break;
            // hand the stream position back to the uniform reader (block headers, stored blocks)
            __syncthreads();
            in_pos = (p >> 5) << 2;
            bb = 0; nb = 0;
            refill();
            { const int r = (int)(p & 31u); bb >>= r; nb -= r; }
        } else {
            for (;;) {
                refill();
                uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.lit[(uint32_t)bb & ((1u << kLitRoot) - 1u)]);
                if (ent_is_ptr(e)) {
                    const uint32_t sb = ent_bits(e);
                    e = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.lit[ent_ptr_start(e) + (((uint32_t)(bb >> kLitRoot)) & ((1u << sb) - 1u))]);
                    bb >>= kLitRoot; nb -= kLitRoot;
                }
                const uint32_t nbits = ent_bits(e);
                if (nbits == 0u) { err = kInfBadSymbol; break; }
                bb >>= nbits; nb -= (int)nbits;
                if (!(e & (kEntLen | kEntSpecial))) {     // literal
                    if (pos >= ulen) { err = kInfOverrun; break; }
                    if (lane == 0) sh.win[pos & (kWinBytes - 1)] = (uint8_t)ent_v8(e);
                    pos += 1u;
                } else if (!(e & kEntLen)) {              // end of block
                    if (e & kEntBadBit) err = kInfBadSymbol;
                    break;
                } else {                                  // length + distance
                    refill();
                    const uint32_t len = 3u + ent_v8(e) + take((int)((e >> 12) & 7u));
                    refill();
                    uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.dist[(uint32_t)bb & ((1u << kDistRoot) - 1u)]);
                    if (d & kEntPtr) {
                        const uint32_t sb = ent_bits(d);
                        d = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.dist[ent_ptr_start(d) + (((uint32_t)(bb >> kDistRoot)) & ((1u << sb) - 1u))]);
                        bb >>= kDistRoot; nb -= kDistRoot;
                    }
                    const uint32_t dbits = ent_bits(d);
                    if (dbits == 0u || (d & (kEntPtr | kEntSpecial)) != 0u) { err = kInfBadSymbol; break; }
                    bb >>= dbits; nb -= (int)dbits;
                    refill();
                    const uint32_t dxb = (d >> 4) & 15u;
                    const uint32_t dist = 1u + (((d >> 8) & 3u) << dxb) + take((int)dxb);
                    if (dist > pos) { err = kInfBadDistance; break; }
                    if (pos + len > ulen) { err = kInfOverrun; break; }
                    __syncthreads();
                    if (dist <= (uint32_t)kWinBytes - 258u) {
                        // copy inside the LDS window: byte k comes from `dist` back; where the match overlaps itself the pattern repeats
                        for (uint32_t k = (uint32_t)lane; k < len; k += 64u) {
                            const uint32_t from = dist >= len ? pos - dist + k : pos - dist + (k % dist);
                            sh.win[(pos + k) & (kWinBytes - 1)] = sh.win[from & (kWinBytes - 1)];
                        }
                    } else {
                        // the source lies behind the LDS window: it has been flushed (whatever is older than the window minus a
                        // flush piece has), so read it back from the member's output -- once the stores have landed, and past
                        // this CU's L1, which may hold an older state of the line (dist > len here: no self-overlap)
                        __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) expcnt(0) lgkmcnt(0)
                        for (uint32_t k = (uint32_t)lane; k < len; k += 64u)
                            sh.win[(pos + k) & (kWinBytes - 1)] = __hip_atomic_load(dst + (pos - dist + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    __syncthreads();
                    pos += len;
                }
                if (pos - flushed >= (uint32_t)kFlush + 272u) flush_to((flushed + kFlush) & ~15u);   // (unflushed < a piece + two matches: the window keeps the rest as history)
            }
        }
    }
    if (err == kInfOk && pos != ulen) err = kInfShort;
    if (err == kInfOk) flush_to(ulen);
    // CRC-32 of the member is checked by k_bgzf_crc (below) on the inflated bytes in HBM
    if (lane == 0) status[m] = (uint32_t)err;
}

// CRC-32 of every member's payload (what bgzf.c verifies, kent/src/htslib/bgzf.c:421-530): one wave per member,
// 64 contiguous slices, combined in order.  crc(A || B) = shift(crc(A), |B|) ^ crc(B) with the shift by a fixed slice
// length applied through four 256-entry tables (computed by the host: `shift_tab[4][256]` for slices of kCrcSlice bytes).
constexpr int kCrcSlice = 1056;   // 63 slices + a head cover the 64 KiB a member can hold
__global__ __launch_bounds__(64) void k_bgzf_crc(const uint8_t *__restrict__ out, const Member *__restrict__ members, int first_member, int nmembers,
                                                 const uint32_t *__restrict__ crc_tab, const uint32_t *__restrict__ shift_tab,
                                                 uint32_t *__restrict__ status) {
    __shared__ uint32_t tab[256];
    __shared__ uint32_t part[64];
    const int m = first_member + (int)blockIdx.x;   // (the members of one upload piece, behind their inflate launch on the same stream)
    if (m >= nmembers) return;
    const Member mb = members[m];
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < 256; i += 64) tab[i] = crc_tab[i];
    __syncthreads();
    // slice 0 takes the odd head, slices 1.. are kCrcSlice bytes each
    const uint32_t n = mb.ulen;
    const uint32_t nfull = n / kCrcSlice, head = n - nfull * kCrcSlice;
    const uint8_t *p = out + mb.uoff;
    uint32_t crc = 0u;
    // lane l handles slice l: slice 0 = [0, head) (starts from the all-ones register), slice k >= 1 = [head + (k-1) S, head + k S)
    // (from a zero register: the CRC of a message part under the linear part of the recurrence)
    if (lane == 0) {
        crc = 0xffffffffu;
        for (uint32_t i = 0; i < head; ++i) crc = crc_byte(tab, crc, p[i]);
    } else if ((uint32_t)lane <= nfull) {
        const uint8_t *q = p + head + (size_t)(lane - 1) * kCrcSlice;
        for (int i = 0; i < kCrcSlice; ++i) crc = crc_byte(tab, crc, q[i]);
    }
    part[lane] = crc;
    __syncthreads();
    if (lane == 0) {
        uint32_t c = part[0];
        for (uint32_t k = 1; k <= nfull && k < 64; ++k) {
            // advance c over kCrcSlice zero bytes, then add the slice's own remainder
            c = shift_tab[c & 0xffu] ^ shift_tab[256 + ((c >> 8) & 0xffu)] ^ shift_tab[512 + ((c >> 16) & 0xffu)] ^ shift_tab[768 + (c >> 24)];
            c ^= part[k];
        }
        c ^= 0xffffffffu;
        if (status[m] == 0u && c != mb.crc) status[m] = (uint32_t)kInfCrc;
    }
}

// ---------------------------------------------------------------- BAM records
// Does a BAM record plausibly start at `q` (bytes available: avail)?  The checks the host's guess makes
// (bam_stager.cpp guess_record_start): a length prefix that holds the fixed fields, name, CIGAR, sequence and
// qualities; a reference id in range; a position that is a position.
__device__ __forceinline__ uint32_t ld32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
__device__ __forceinline__ uint32_t ld16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

__device__ __forceinline__ bool plausible_record(const uint8_t *q, uint64_t avail, uint32_t n_ref) {
    if (avail < 36) return false;
    const uint32_t bs = ld32(q);
    if (bs < 32 || bs > (1u << 26)) return false;
    const int32_t tid = (int32_t)ld32(q + 4), pos = (int32_t)ld32(q + 8);
    if (tid < -1 || tid >= (int32_t)n_ref || pos < -1) return false;
    const uint32_t l_name = q[12], n_cig = ld16(q + 16), l_seq = ld32(q + 20);
    if (l_name == 0 || l_seq > (1u << 26)) return false;
    const int32_t ntid = (int32_t)ld32(q + 24), npos = (int32_t)ld32(q + 28);
    if (ntid < -1 || ntid >= (int32_t)n_ref || npos < -1) return false;
    return (uint64_t)32 + l_name + 4ull * n_cig + (l_seq + 1) / 2 + l_seq <= bs;
}

constexpr int kMaxRecPerMember = 1824;   // 65536 / 36 + 3: a record takes at least 36 bytes of the stream
constexpr int kGuessChain = 3;           // records in a row that have to look like records

struct MemberChain {
    uint64_t first;      // stream offset of the first record start at or behind the member's begin (guessed or given)
    uint64_t next;       // where the chain from `first` leaves the member: the first record start at or behind its end
    uint32_t nrec;       // record starts in [first, member end)
    uint32_t flags;      // 1: no plausible start found, 2: a length prefix below the fixed fields met on the way
};

// One wave per member.  forced[m] != ~0: start there instead of guessing (the host found that the preceding member's
// chain ends there).  rec_off[m * kMaxRecPerMember + k] = offset of record k relative to the member's begin.
__global__ __launch_bounds__(64) void k_bam_chain(const uint8_t *__restrict__ stream, uint64_t stream_len, const Member *__restrict__ members,
                                                  int nmembers, int member_lo, uint32_t n_ref, uint64_t first_record, const uint64_t *__restrict__ forced,
                                                  MemberChain *chain, uint32_t *rec_off, uint64_t stop_at) {
    const int m = member_lo + (int)blockIdx.x;
    if (m >= nmembers) return;
    const int lane = threadIdx.x & 63;
    const Member mb = members[m];
    // (region reads: no record is looked for at or behind `stop_at`, the end of the last chunk of the index)
    const uint64_t begin = mb.uoff, end = mb.uoff + mb.ulen < stop_at ? mb.uoff + mb.ulen : stop_at;
    MemberChain mc;
    mc.first = ~0ull; mc.next = ~0ull; mc.nrec = 0; mc.flags = 0;
    if (begin >= end) {   // wholly behind the stop: nothing starts here
        if (lane == 0) { mc.first = stop_at; mc.next = stop_at; chain[m] = mc; }
        return;
    }
    uint64_t start = forced[m];
    if (start == ~0ull && first_record >= begin && first_record < end) start = first_record;   // the member that holds the end of the header
    if (start == ~0ull && end <= first_record) {   // header only: nothing starts here
        if (lane == 0) { mc.first = first_record; mc.next = first_record; chain[m] = mc; }
        return;
    }
    if (start == ~0ull) {
        // guess: lanes try consecutive offsets, 64 at a time; the first offset from which kGuessChain records chain
        for (uint64_t base = begin; base < end && start == ~0ull; base += 64) {
            const uint64_t o = base + (uint64_t)lane;
            bool ok = o < end;
            uint64_t q = o;
            for (int k = 0; k < kGuessChain && ok; ++k) {
                if (q >= stream_len) break;                      // chained to the end of the stream: fine
                ok = plausible_record(stream + q, stream_len - q, n_ref) && q + 4 + ld32(stream + q) <= stream_len;
                if (ok) q += 4 + (uint64_t)ld32(stream + q);
            }
            const unsigned long long hit = __ballot(ok);
            if (hit) start = base + (uint64_t)__builtin_ctzll(hit);
        }
        if (start == ~0ull) {   // no record starts in this member (one record spans it)
            if (lane == 0) { mc.flags = 1u; chain[m] = mc; }
            return;
        }
    }
    // walk the chain (uniform)
    uint64_t q = start;
    uint32_t n = 0;
    while (q < end) {
        if (q + 4 > stream_len) { mc.flags |= 2u; break; }   // a length prefix cut by the end of the stream
        const uint32_t bs = ld32(stream + q);
        if (n < (uint32_t)kMaxRecPerMember && lane == 0) rec_off[(size_t)m * kMaxRecPerMember + n] = (uint32_t)(q - begin);
        ++n;
        if (bs < 32) { mc.flags |= 2u; break; }              // (the record decode reports it)
        q += 4 + (uint64_t)bs;
    }
    if (lane == 0) { mc.first = start; mc.next = q; mc.nrec = n; chain[m] = mc; }
}

// error codes of the record decode, in the order the host's serial walk checks them (bam_stager.cpp decode_span_cols)
enum { kRecOk = 0, kRecTruncated = 1, kRecBadSize = 2, kRecTidRange = 3, kRecNegPos = 4, kRecUnsorted = 5, kRecCigarOverrun = 6,
       kRecUnknownOp = 7, kRecEndBeyond = 8, kRecTooLong = 9, kRecDeletionOrder = 10 };

struct RecOut {
    int32_t tid, spos, pos;     // reference id, first aligned position (the sort key of the packed format), POS field
    uint32_t L;                 // aligned positions
    uint32_t nruns;             // maximal runs of aligned positions
    uint16_t flag;
    uint8_t err;
    uint8_t placed;             // 0 unplaced (tid < 0), 1 placed, 2 placed but outside every requested region (region reads: not staged)
    int32_t lseq;               // l_seq (pysam's query_length)
    uint32_t mapq;              // bits 0-7 MAPQ; bits 16-31 the NH:i tag (clamped to 65 535; 0: the record has none)
    int32_t end;                // one past the last aligned position (spos + 1 without aligned bases): the overlap test of region reads
};

// NH:i of a record (bam_stager.cpp aux_nh, bit for bit): the auxiliary fields behind the qualities, {tag[2], type, value}
// each (SAM spec 4.2.4; htslib bam_aux_get's walk); the value of an integer-typed NH clamped to [0, 65 535], 0 without one.
__device__ __forceinline__ uint32_t aux_nh(const uint8_t *after_cigar, int32_t l_seq, const uint8_t *end) {
    if (l_seq < 0) return 0u;
    const uint8_t *a = after_cigar + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
    while (a + 3 <= end && a >= after_cigar) {
        const uint32_t t0 = a[0], t1 = a[1], type = a[2];
        a += 3;
        size_t sz;
        if (type == 'A' || type == 'c' || type == 'C') sz = 1;
        else if (type == 's' || type == 'S') sz = 2;
        else if (type == 'i' || type == 'I' || type == 'f') sz = 4;
        else if (type == 'Z' || type == 'H') {
            const uint8_t *z = a;
            while (z < end && *z) ++z;
            if (z >= end) return 0u;
            sz = (size_t)(z - a) + 1;
        } else if (type == 'B') {
            if (a + 5 > end) return 0u;
            const uint32_t sub = a[0];
            const uint32_t cnt = (uint32_t)a[1] | ((uint32_t)a[2] << 8) | ((uint32_t)a[3] << 16) | ((uint32_t)a[4] << 24);
            const size_t es = (sub == 'c' || sub == 'C') ? 1 : ((sub == 's' || sub == 'S') ? 2 : ((sub == 'i' || sub == 'I' || sub == 'f') ? 4 : 0));
            if (!es) return 0u;
            sz = 5 + (size_t)cnt * es;
        } else return 0u;
        if ((size_t)(end - a) < sz) return 0u;
        if (t0 == 'N' && t1 == 'H') {
            long long v;
            if (type == 'c') v = (int8_t)a[0];
            else if (type == 'C') v = a[0];
            else if (type == 's') v = (int16_t)((uint32_t)a[0] | ((uint32_t)a[1] << 8));
            else if (type == 'S') v = (uint32_t)a[0] | ((uint32_t)a[1] << 8);
            else if (type == 'i') v = (int32_t)((uint32_t)a[0] | ((uint32_t)a[1] << 8) | ((uint32_t)a[2] << 16) | ((uint32_t)a[3] << 24));
            else if (type == 'I') v = (uint32_t)a[0] | ((uint32_t)a[1] << 8) | ((uint32_t)a[2] << 16) | ((uint32_t)a[3] << 24);
            else return 0u;
            return (uint32_t)(v < 0 ? 0 : (v > 65535 ? 65535 : v));
        }
        a += sz;
    }
    return 0u;
}

// One thread per record start (member-major, k_bam_chain's order): fields and CIGAR -> RecOut.
__global__ __launch_bounds__(256) void k_bam_fields(const uint8_t *__restrict__ stream, uint64_t stream_len, const Member *__restrict__ members,
                                                    const uint64_t *__restrict__ rec_base, const MemberChain *__restrict__ chain,
                                                    const uint32_t *__restrict__ rec_off, int nmembers, int64_t nrec, uint32_t n_ref,
                                                    const uint32_t *__restrict__ rec_member, RecOut *recs) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nrec) return;
    // member of record i: rec_member holds it for every 256-record group start; walk forward from there
    int m = (int)rec_member[i >> 8];
    while (m + 1 < nmembers && (int64_t)rec_base[m + 1] <= i) ++m;
    const uint64_t q = members[m].uoff + rec_off[(size_t)m * kMaxRecPerMember + (size_t)(i - (int64_t)rec_base[m])];
    RecOut o;
    o.tid = -1; o.spos = 0; o.pos = 0; o.L = 0; o.nruns = 0; o.flag = 0; o.err = kRecOk; o.placed = 0; o.lseq = 0; o.mapq = 0; o.end = 0;
    if (q + 4 > stream_len) { o.err = kRecTruncated; recs[i] = o; return; }
    const uint32_t bs = ld32(stream + q);
    if (bs < 32) { o.err = kRecBadSize; recs[i] = o; return; }
    if (q + 4 + (uint64_t)bs > stream_len) { o.err = kRecTruncated; recs[i] = o; return; }
    const uint8_t *r = stream + q + 4;
    const int32_t tid = (int32_t)ld32(r), pos = (int32_t)ld32(r + 4);
    const uint32_t l_name = r[8], n_cig = ld16(r + 12);
    o.flag = (uint16_t)ld16(r + 14);
    o.mapq = r[9];
    o.lseq = (int32_t)ld32(r + 16);
    o.tid = tid; o.pos = pos; o.spos = pos;
    if (tid < 0) { recs[i] = o; return; }                   // unplaced: counted, not staged
    o.placed = 1;
    if (tid >= (int32_t)n_ref) { o.err = kRecTidRange; recs[i] = o; return; }
    if (pos < 0) { o.err = kRecNegPos; recs[i] = o; return; }
    if ((uint64_t)32 + l_name + 4ull * n_cig > bs) { o.err = kRecCigarOverrun; recs[i] = o; return; }
    const uint8_t *cig = r + 32 + l_name;
    o.mapq |= aux_nh(cig + 4ull * n_cig, o.lseq, r + bs) << 16;
    int64_t ref = pos, L = 0, run_end = -1;
    int32_t first_run = -1;
    uint32_t nruns = 0;
    for (uint32_t c = 0; c < n_cig; ++c) {
        const uint32_t v = ld32(cig + 4 * c), op = v & 15u, len = v >> 4;
        if (op == 0u || op == 7u || op == 8u) {             // M = X: aligned positions
            if (len) {
                if (run_end != ref) { ++nruns; if (first_run < 0) first_run = (int32_t)ref; }
                ref += len; L += len; run_end = ref;
            }
        } else if (op == 2u || op == 3u) ref += len;        // D N: reference only
        else if (op == 1u || op == 4u || op == 5u || op == 6u) {}   // I S H P
        else { o.err = kRecUnknownOp; break; }
    }
    if (o.err == kRecOk && ref > 0x7fffffffLL) o.err = kRecEndBeyond;
    if (o.err == kRecOk && L > 0x7fffffffLL) o.err = kRecTooLong;
    o.L = (uint32_t)L; o.nruns = nruns;
    if (nruns) o.spos = first_run;
    o.end = (int32_t)(nruns ? (run_end > 0x7fffffffLL ? 0x7fffffffLL : run_end) : (int64_t)pos + 1);
    recs[i] = o;
}

// order checks between neighbours (the host's serial walk: sorted by (tid, POS), unplaced reads last, and the first
// aligned positions in order too) + the lowest record index with a defect
__global__ __launch_bounds__(256) void k_bam_order(const RecOut *__restrict__ recs, int64_t nrec, const uint32_t *__restrict__ placed_before,
                                                   unsigned long long *first_err) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nrec) return;
    const RecOut o = recs[i];
    uint32_t err = o.err;
    if (o.placed && (err == kRecOk || err >= kRecCigarOverrun)) {
        // previous PLACED record (placed records are contiguous in a valid file: an unplaced one in between is the defect)
        if (i > 0) {
            const RecOut pr = recs[i - 1];
            uint32_t oerr = 0;
            if (!pr.placed) oerr = kRecUnsorted;                               // a placed record behind an unplaced one
            else if (o.tid < pr.tid || (o.tid == pr.tid && o.pos < pr.pos)) oerr = kRecUnsorted;
            else if (pr.err == kRecOk && o.tid == pr.tid && pr.spos > o.spos && err == kRecOk) oerr = kRecDeletionOrder;
            if (oerr == kRecUnsorted) err = kRecUnsorted;                      // checked before the record's own CIGAR
            else if (oerr && err == kRecOk) err = oerr;
        }
    }
    if (err) atomicMin(first_err, ((unsigned long long)i << 8) | (unsigned long long)err);
}

// columns of the staged (placed) records at their scanned indices; runs of the multi-run records at theirs
__global__ __launch_bounds__(256) void k_bam_columns(const uint8_t *__restrict__ stream, const Member *__restrict__ members,
                                                     const uint64_t *__restrict__ rec_base, const uint32_t *__restrict__ rec_off, int nmembers,
                                                     const uint32_t *__restrict__ rec_member, const RecOut *__restrict__ recs, int64_t nrec,
                                                     const uint32_t *__restrict__ staged_at, const uint32_t *__restrict__ run_at,
                                                     int32_t *tid, int32_t *pos, uint16_t *alen, uint8_t *flags, uint8_t *nblk,
                                                     int32_t *blk_start, int32_t *blk_len, uint32_t *wide_flag,
                                                     uint16_t *flag16, uint8_t *mapq, int32_t *lseq, uint16_t *nh) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nrec) return;
    const RecOut o = recs[i];
    if (o.placed != 1) return;
    const uint32_t k = staged_at[i];
    const bool wide = o.L > 65535u || o.nruns > 255u || (o.L == 65535u && o.nruns == 255u);
    tid[k] = o.tid;
    pos[k] = o.spos;
    alen[k] = wide ? (uint16_t)65535 : (uint16_t)o.L;
    flags[k] = (o.flag & 0x10) ? 1 : 0;
    nblk[k] = wide ? (uint8_t)255 : (uint8_t)o.nruns;
    flag16[k] = o.flag;
    mapq[k] = (uint8_t)o.mapq;
    nh[k] = (uint16_t)(o.mapq >> 16);
    lseq[k] = o.lseq;
    if (wide) wide_flag[k] = 1u;
    if (o.nruns < 2u) return;
    int m = (int)rec_member[i >> 8];
    while (m + 1 < nmembers && (int64_t)rec_base[m + 1] <= i) ++m;
    const uint8_t *r = stream + members[m].uoff + rec_off[(size_t)m * kMaxRecPerMember + (size_t)(i - (int64_t)rec_base[m])] + 4;
    const uint32_t l_name = r[8], n_cig = ld16(r + 12);
    const uint8_t *cig = r + 32 + l_name;
    int64_t ref = o.pos, run_end = -1;
    int64_t w = (int64_t)run_at[i] - 1;
    for (uint32_t c = 0; c < n_cig; ++c) {
        const uint32_t v = ld32(cig + 4 * c), op = v & 15u, len = v >> 4;
        if (op == 0u || op == 7u || op == 8u) {
            if (len) {
                if (run_end != ref) { ++w; blk_start[w] = (int32_t)ref; blk_len[w] = 0; }
                blk_len[w] += (int32_t)len;
                ref += len; run_end = ref;
            }
        } else if (op == 2u || op == 3u) ref += len;
    }
}

// Region reads: a placed record stays iff it overlaps one of the requested regions (ascending, merged, by reference
// id) -- htslib's test, pos < end && endpos > beg (hts.c:1924-1960), on the first aligned position and the end of the
// last aligned run as the host reader applies it (bam_stager.cpp decode_regions).  Runs behind the order checks.
__global__ __launch_bounds__(256) void k_bam_region_filter(RecOut *recs, int64_t nrec, int nreg, const int32_t *__restrict__ reg_tid,
                                                           const int64_t *__restrict__ reg_beg, const int64_t *__restrict__ reg_end) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nrec) return;
    RecOut o = recs[i];
    if (o.placed != 1) return;
    // first region (ordered by reference id, then start) whose end lies beyond the record's first position
    int lo = 0, hi = nreg;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (reg_tid[mid] < o.tid || (reg_tid[mid] == o.tid && reg_end[mid] <= (int64_t)o.spos)) lo = mid + 1; else hi = mid;
    }
    const bool keep = lo < nreg && reg_tid[lo] == o.tid && reg_beg[lo] < (int64_t)o.end;
    if (!keep) { o.placed = 2; recs[i] = o; }
}

// per-record scan inputs: staged (placed) flag and the runs a multi-run record keeps
constexpr int kScanInputsPerThread = 8;   // records per thread of k_bam_scan_inputs (one atomic per counter and workgroup: a single hot address takes ~90 atomics per us)
__global__ __launch_bounds__(256) void k_bam_scan_inputs(const RecOut *__restrict__ recs, int64_t nrec, uint32_t *placed, uint32_t *runs,
                                                         unsigned long long *counts) {
    // counts[0] mapped (flag 0x4 unset), counts[1] unplaced
    unsigned long long a = 0, b = 0;
#pragma unroll
    for (int k = 0; k < kScanInputsPerThread; ++k) {
        const int64_t i = ((int64_t)blockIdx.x * kScanInputsPerThread + k) * 256 + threadIdx.x;
        if (i < nrec) {
            const RecOut o = recs[i];
            placed[i] = o.placed == 1 ? 1u : 0u;
            runs[i] = (o.placed == 1 && o.nruns >= 2u) ? o.nruns : 0u;
            a += (o.placed == 2 || (o.flag & 0x4)) ? 0u : 1u;   // (records outside the requested regions count for nothing)
            b += o.placed ? 0u : 1u;
        }
    }
    for (int o2 = 32; o2 > 0; o2 >>= 1) { a += __shfl_down(a, o2, 64); b += __shfl_down(b, o2, 64); }
    __shared__ unsigned long long s_a[4], s_b[4];
    if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = a; s_b[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = s_a[0] + s_a[1] + s_a[2] + s_a[3];
        b = s_b[0] + s_b[1] + s_b[2] + s_b[3];
        if (a) atomicAdd(&counts[0], a);
        if (b) atomicAdd(&counts[1], b);
    }
}

} // namespace pcbam
