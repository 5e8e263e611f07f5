// bam_stager.cpp -- native BAM -> packed-array stager (host side, no GPU code).
//
// The step *before* the counting path: the reference gets its reads from pysam/htslib
// (`pysam.AlignmentFile(X, "rb")`, `.fetch`, `read.positions`, `read.is_reverse`, `.mapped`;
// plastid/genomics/genome_array.py:660, 669, 690, 800-815).  pysam is not part of this
// product; this file reads a coordinate-sorted BAM (SAM/BAM spec v1: BGZF blocks, BAM
// records) straight into the flat arrays `pc_add_alignment_file` stages to HBM:
//   tid, pos, alen (= number of M/=/X reference positions), flags (bit0 = reverse strand),
//   nblk (maximal runs of contiguous aligned positions), runs of the gapped records.
// BGZF members are independent, so they are inflated by a pool of threads.
//
// C ABI (ctypes: plastid_amd/bam.py):
//   pb_open / pb_close, pb_nref / pb_ref_name / pb_ref_length,
//   pb_load  (decode the whole file), pb_counts, pb_fill (copy into caller arrays)
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(const std::string &m) {
    g_err = m;
    return -1;
}

struct Block {
    size_t coff;   // offset of the gzip member in the file
    uint32_t clen; // compressed member length
    uint32_t ulen; // uncompressed length (ISIZE)
    size_t uoff;   // offset in the inflated stream
};

struct Bam {
    std::string path;
    std::vector<std::string> ref_names;
    std::vector<int32_t> ref_lengths;
    // decoded records
    std::vector<int32_t> tid, pos, blk_start, blk_len;
    std::vector<uint16_t> alen;
    std::vector<uint8_t> flags, nblk;
    int64_t mapped = 0, unplaced = 0, total = 0;
    bool loaded = false;
};

bool read_file(const std::string &path, std::vector<uint8_t> &buf) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize((size_t)n);
    size_t got = n > 0 ? fread(buf.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    return got == (size_t)n;
}

inline uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// index the BGZF members (RFC 1952 gzip header with the 'BC' extra subfield carrying BSIZE)
int scan_blocks(const std::vector<uint8_t> &file, std::vector<Block> &blocks, size_t &total_u) {
    size_t off = 0;
    total_u = 0;
    while (off < file.size()) {
        if (off + 18 > file.size()) return fail("truncated BGZF header");
        const uint8_t *h = file.data() + off;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return fail("not a BGZF file (bad gzip member header)");
        const uint16_t xlen = rd16(h + 10);
        if (off + 12 + xlen > file.size()) return fail("truncated BGZF extra field");
        int bsize = -1;
        for (size_t x = 0; x + 4 <= xlen;) {
            const uint8_t *sf = h + 12 + x;
            const uint16_t slen = rd16(sf + 2);
            if (sf[0] == 'B' && sf[1] == 'C' && slen == 2) bsize = rd16(sf + 4);
            x += 4 + slen;
        }
        if (bsize < 0) return fail("BGZF member without BC subfield");
        const size_t clen = (size_t)bsize + 1;
        if (off + clen > file.size()) return fail("truncated BGZF member");
        const uint32_t isize = rd32(file.data() + off + clen - 4);
        blocks.push_back({off, (uint32_t)clen, isize, total_u});
        total_u += isize;
        off += clen;
    }
    return 0;
}

int inflate_block(const std::vector<uint8_t> &file, const Block &b, uint8_t *dst) {
    if (b.ulen == 0) return 0;
    const uint8_t *h = file.data() + b.coff;
    const size_t hdr = 12 + rd16(h + 10);
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) return -1;
    zs.next_in = const_cast<uint8_t *>(h + hdr);
    zs.avail_in = (uInt)(b.clen - hdr - 8);
    zs.next_out = dst;
    zs.avail_out = b.ulen;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    if (rc != Z_STREAM_END || zs.avail_out != 0) return -1;
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), dst, b.ulen);
    if (crc != rd32(h + b.clen - 8)) return -2;
    return 0;
}

int decode(Bam &bam, int nthreads) {
    std::vector<uint8_t> file;
    if (!read_file(bam.path, file)) return fail("cannot read " + bam.path);
    std::vector<Block> blocks;
    size_t total_u = 0;
    if (scan_blocks(file, blocks, total_u) != 0) return -1;
    std::vector<uint8_t> data(total_u);
    // inflate all members in parallel
    std::atomic<size_t> next(0);
    std::atomic<int> bad(0);
    if (nthreads < 1) nthreads = 1;
    nthreads = (int)std::min<size_t>((size_t)nthreads, std::max<size_t>(blocks.size(), 1));
    auto worker = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= blocks.size()) return;
            const int rc = inflate_block(file, blocks[i], data.data() + blocks[i].uoff);
            if (rc != 0) bad.store(rc);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; ++t) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    if (bad.load() == -2) return fail("BGZF CRC mismatch in " + bam.path);
    if (bad.load() != 0) return fail("BGZF inflate failed in " + bam.path);
    file.clear();
    file.shrink_to_fit();

    // ---- BAM header
    const uint8_t *p = data.data(), *end = data.data() + data.size();
    if (end - p < 12 || std::memcmp(p, "BAM\1", 4) != 0) return fail("not a BAM file (bad magic)");
    const uint32_t l_text = rd32(p + 4);
    p += 8;
    if ((size_t)(end - p) < (size_t)l_text + 4) return fail("truncated BAM header");
    p += l_text;
    const uint32_t n_ref = rd32(p);
    p += 4;
    bam.ref_names.clear();
    bam.ref_lengths.clear();
    for (uint32_t r = 0; r < n_ref; ++r) {
        if (end - p < 4) return fail("truncated BAM reference list");
        const uint32_t l_name = rd32(p);
        p += 4;
        if ((size_t)(end - p) < (size_t)l_name + 4) return fail("truncated BAM reference list");
        bam.ref_names.emplace_back(reinterpret_cast<const char *>(p), l_name ? l_name - 1 : 0);
        p += l_name;
        bam.ref_lengths.push_back((int32_t)rd32(p));
        p += 4;
    }

    // ---- alignment records
    int32_t last_tid = -1, last_pos = -1;
    bool seen_unplaced = false;
    std::vector<std::pair<int32_t, int32_t>> runs;
    while (p < end) {
        if (end - p < 4) return fail("truncated BAM record");
        const uint32_t block_size = rd32(p);
        p += 4;
        if ((size_t)(end - p) < block_size || block_size < 32) return fail("truncated BAM record");
        const uint8_t *r = p;
        p += block_size;
        const int32_t tid = (int32_t)rd32(r), pos = (int32_t)rd32(r + 4);
        const uint8_t l_read_name = r[8];
        const uint16_t n_cigar = rd16(r + 12), flag = rd16(r + 14);
        bam.total += 1;
        if (!(flag & 0x4)) bam.mapped += 1;
        if (tid < 0) { // unplaced reads sit at the end of a sorted BAM; fetch() never returns them
            bam.unplaced += 1;
            seen_unplaced = true;
            continue;
        }
        if (tid >= (int32_t)n_ref) return fail("BAM record with reference id out of range");
        if (seen_unplaced || tid < last_tid || (tid == last_tid && pos < last_pos))
            return fail("BAM file is not coordinate sorted: " + bam.path);
        last_tid = tid;
        last_pos = pos;
        if ((size_t)32 + l_read_name + (size_t)n_cigar * 4 > block_size) return fail("corrupt BAM record (cigar overruns block)");
        const uint8_t *cig = r + 32 + l_read_name;
        runs.clear();
        int64_t ref = pos, L = 0;
        for (uint16_t c = 0; c < n_cigar; ++c) {
            const uint32_t v = rd32(cig + 4 * c);
            const uint32_t op = v & 0xf, len = v >> 4;
            switch (op) {
            case 0: case 7: case 8: // M = X : aligned positions
                if (len) {
                    if (!runs.empty() && (int64_t)runs.back().first + runs.back().second == ref) runs.back().second += (int32_t)len;
                    else runs.emplace_back((int32_t)ref, (int32_t)len);
                    ref += len;
                    L += len;
                }
                break;
            case 2: case 3: // D N : reference only
                ref += len;
                break;
            case 1: case 4: case 5: case 6: // I S H P
                break;
            default:
                return fail("unknown CIGAR operation in " + bam.path);
            }
        }
        if (L > 65535) return fail("alignment with more than 65535 aligned positions is not supported");
        if (runs.size() > 255) return fail("alignment with more than 255 aligned runs is not supported");
        // the packed format keys a record on its first aligned position; a CIGAR that opens with
        // D/N (not produced by aligners) is accepted only if that keeps the file order
        const int32_t spos = runs.empty() ? pos : runs[0].first;
        if (!bam.tid.empty() && bam.tid.back() == tid && bam.pos.back() > spos)
            return fail("alignment starting with a deletion breaks coordinate order; not supported");
        bam.tid.push_back(tid);
        bam.pos.push_back(spos);
        bam.alen.push_back((uint16_t)L);
        bam.flags.push_back((flag & 0x10) ? 1 : 0);
        bam.nblk.push_back((uint8_t)runs.size());
        if (runs.size() >= 2)
            for (auto &x : runs) {
                bam.blk_start.push_back(x.first);
                bam.blk_len.push_back(x.second);
            }
    }
    bam.loaded = true;
    return 0;
}

} // namespace

extern "C" {

const char *pb_last_error(void) { return g_err.c_str(); }

void *pb_open(const char *path) {
    if (!path) {
        fail("pb_open: NULL path");
        return nullptr;
    }
    FILE *f = fopen(path, "rb");
    if (!f) {
        fail(std::string("cannot open ") + path);
        return nullptr;
    }
    fclose(f);
    Bam *b = new Bam();
    b->path = path;
    return b;
}

void pb_close(void *h) { delete static_cast<Bam *>(h); }

// decode the whole file with `nthreads` inflate threads (<= 0: hardware concurrency)
int pb_load(void *h, int nthreads) {
    Bam *b = static_cast<Bam *>(h);
    if (!b) return fail("pb_load: NULL handle");
    if (b->loaded) return 0;
    if (nthreads <= 0) nthreads = (int)std::max(1u, std::thread::hardware_concurrency());
    return decode(*b, nthreads);
}

int pb_nref(void *h) { return h ? (int)static_cast<Bam *>(h)->ref_names.size() : -1; }
const char *pb_ref_name(void *h, int i) { return static_cast<Bam *>(h)->ref_names[(size_t)i].c_str(); }
int32_t pb_ref_length(void *h, int i) { return static_cast<Bam *>(h)->ref_lengths[(size_t)i]; }

// counts[0] = staged (placed) records, [1] = runs of gapped records, [2] = mapped reads
// (flag 0x4 unset, what pysam's AlignmentFile.mapped reports), [3] = all records
int pb_counts(void *h, int64_t *counts) {
    Bam *b = static_cast<Bam *>(h);
    if (!b || !b->loaded) return fail("pb_counts: file not loaded");
    counts[0] = (int64_t)b->tid.size();
    counts[1] = (int64_t)b->blk_start.size();
    counts[2] = b->mapped;
    counts[3] = b->total;
    return 0;
}

int pb_fill(void *h, int32_t *tid, int32_t *pos, uint16_t *alen, uint8_t *flags, uint8_t *nblk, int32_t *blk_start,
            int32_t *blk_len) {
    Bam *b = static_cast<Bam *>(h);
    if (!b || !b->loaded) return fail("pb_fill: file not loaded");
    const size_t n = b->tid.size(), m = b->blk_start.size();
    if (n) {
        std::memcpy(tid, b->tid.data(), n * 4);
        std::memcpy(pos, b->pos.data(), n * 4);
        std::memcpy(alen, b->alen.data(), n * 2);
        std::memcpy(flags, b->flags.data(), n);
        std::memcpy(nblk, b->nblk.data(), n);
    }
    if (m) {
        std::memcpy(blk_start, b->blk_start.data(), m * 4);
        std::memcpy(blk_len, b->blk_len.data(), m * 4);
    }
    return 0;
}

} // extern "C"
