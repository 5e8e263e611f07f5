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
#include <chrono>
#include <thread>
#include <sys/mman.h>
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

// Records decoded from one contiguous piece of the inflated stream (the pieces are decoded in
// parallel and stitched in file order).
struct Part {
    std::vector<int32_t> tid, pos, blk_start, blk_len;
    std::vector<uint16_t> alen;
    std::vector<uint8_t> flags, nblk;
    int64_t mapped = 0, unplaced = 0, total = 0;
    // first defect found inside the piece: global record index, message; `before_order` marks the
    // checks the serial walk makes before it looks at the sort order of a record
    int64_t err_rec = INT64_MAX;
    bool err_before_order = false;
    std::string err;
    // what the stitching needs to know about the piece
    bool any_placed = false, saw_unplaced = false;
    int64_t first_placed_rec = -1;
    int32_t first_tid = -1, first_pos = -1, first_spos = -1;
    int32_t last_tid = -1, last_pos = -1, last_spos = -1;
};

struct Bam {
    std::string path;
    std::vector<std::string> ref_names;
    std::vector<int32_t> ref_lengths;
    std::vector<Part> parts;          // decoded records, in file order
    std::vector<size_t> rec_off, run_off; // where each part starts in the flat arrays
    int64_t mapped = 0, unplaced = 0, total = 0;
    size_t nrec = 0, nrun = 0;
    int threads = 1;
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

// PB_TIMING=1: print the phases of a load to stderr
struct Lap {
    bool on = getenv("PB_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void operator()(const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[bam] %-24s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

int decode(Bam &bam, int nthreads) {
    Lap lap;
    std::vector<uint8_t> file;
    if (!read_file(bam.path, file)) return fail("cannot read " + bam.path);
    std::vector<Block> blocks;
    size_t total_u = 0;
    if (scan_blocks(file, blocks, total_u) != 0) return -1;
    lap("read + index members");
    // the inflated stream: not value-initialised (the inflate threads are the first to touch their
    // members' pages) and on transparent huge pages when large
    struct Raw {
        uint8_t *p = nullptr;
        size_t n = 0;
        explicit Raw(size_t bytes) : n(bytes) {
            const size_t huge = (size_t)2 << 20, want = std::max<size_t>(bytes, 1);
            if (want >= 4 * huge) {
                void *q = nullptr;
                if (posix_memalign(&q, huge, (want + huge - 1) / huge * huge) == 0) {
                    (void)madvise(q, (want + huge - 1) / huge * huge, MADV_HUGEPAGE);
                    p = (uint8_t *)q;
                }
            } else {
                p = (uint8_t *)malloc(want);
            }
        }
        ~Raw() { free(p); }
        uint8_t *data() { return p; }
        size_t size() const { return n; }
    } data(total_u);
    if (!data.p) return fail("out of memory inflating " + bam.path);
    lap("allocate");
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
    lap("inflate");

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

    // ---- alignment records.  A serial walk over the length prefixes cuts the stream into pieces of
    // kPiece records; the pieces are decoded by the thread pool; the stitching pass then applies the
    // checks that span two pieces.  The defect reported is the one of the lowest record index, as
    // in a serial walk (where one record fails two checks, the serial order of the checks decides).
    int64_t kPiece = 1 << 16;
    if (const char *env = getenv("PB_PIECE")) kPiece = std::max(1, atoi(env));   // test knob: tiny pieces exercise the stitching
    std::vector<const uint8_t *> cuts;
    int64_t nwalk = 0;
    bool walk_truncated = false;
    {
        const uint8_t *q = p;
        while (q < end) {
            if (end - q < 4) { walk_truncated = true; break; }
            const uint32_t block_size = rd32(q);
            if ((size_t)(end - q - 4) < block_size || block_size < 32) { walk_truncated = true; break; }
            if (nwalk % kPiece == 0) cuts.push_back(q);
            q += 4 + (size_t)block_size;
            nwalk += 1;
        }
        cuts.push_back(q);
    }
    lap("walk records");
    const size_t nparts = cuts.size() - 1;
    bam.parts.assign(nparts, Part());
    auto decode_piece = [&](size_t k) {
        Part &pt = bam.parts[k];
        const int64_t base = (int64_t)k * kPiece;
        const int64_t count = std::min<int64_t>(kPiece, nwalk - base);
        pt.tid.reserve((size_t)count); pt.pos.reserve((size_t)count); pt.alen.reserve((size_t)count);
        pt.flags.reserve((size_t)count); pt.nblk.reserve((size_t)count);
        const uint8_t *q = cuts[k];
        std::vector<std::pair<int32_t, int32_t>> runs;
        auto bad = [&](int64_t i, bool before_order, const std::string &m) {
            pt.err_rec = base + i; pt.err_before_order = before_order; pt.err = m;
        };
        for (int64_t i = 0; i < count; ++i) {
            const uint32_t block_size = rd32(q);
            const uint8_t *r = q + 4;
            q = r + block_size;
            const int32_t tid = (int32_t)rd32(r), pos = (int32_t)rd32(r + 4);
            const uint8_t l_read_name = r[8];
            const uint16_t n_cigar = rd16(r + 12), flag = rd16(r + 14);
            pt.total += 1;
            if (!(flag & 0x4)) pt.mapped += 1;
            if (tid < 0) { // unplaced reads sit at the end of a sorted BAM; fetch() never returns them
                pt.unplaced += 1;
                pt.saw_unplaced = true;
                continue;
            }
            if (tid >= (int32_t)n_ref) return bad(i, true, "BAM record with reference id out of range");
            if (!pt.any_placed) {     // its order against the previous piece is checked when stitching
                pt.any_placed = true;
                pt.first_placed_rec = base + i;
                pt.first_tid = tid; pt.first_pos = pos;
                if (pt.saw_unplaced) return bad(i, false, "BAM file is not coordinate sorted: " + bam.path);
            } else if (pt.saw_unplaced || tid < pt.last_tid || (tid == pt.last_tid && pos < pt.last_pos)) {
                return bad(i, false, "BAM file is not coordinate sorted: " + bam.path);
            }
            const bool first = pt.tid.empty();
            if ((size_t)32 + l_read_name + (size_t)n_cigar * 4 > block_size) return bad(i, false, "corrupt BAM record (cigar overruns block)");
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
                    return bad(i, false, "unknown CIGAR operation in " + bam.path);
                }
            }
            if (L > 65535) return bad(i, false, "alignment with more than 65535 aligned positions is not supported");
            if (runs.size() > 255) return bad(i, false, "alignment with more than 255 aligned runs is not supported");
            // the packed format keys a record on its first aligned position; a CIGAR that opens with
            // D/N (not produced by aligners) is accepted only if that keeps the file order
            const int32_t spos = runs.empty() ? pos : runs[0].first;
            if (first) pt.first_spos = spos;
            else if (pt.last_tid == tid && pt.last_spos > spos)
                return bad(i, false, "alignment starting with a deletion breaks coordinate order; not supported");
            pt.last_tid = tid; pt.last_pos = pos; pt.last_spos = spos;
            pt.tid.push_back(tid);
            pt.pos.push_back(spos);
            pt.alen.push_back((uint16_t)L);
            pt.flags.push_back((flag & 0x10) ? 1 : 0);
            pt.nblk.push_back((uint8_t)runs.size());
            if (runs.size() >= 2)
                for (auto &x : runs) {
                    pt.blk_start.push_back(x.first);
                    pt.blk_len.push_back(x.second);
                }
        }
    };
    {
        std::atomic<size_t> nextp(0);
        auto pworker = [&]() {
            for (;;) {
                const size_t k = nextp.fetch_add(1);
                if (k >= nparts) return;
                decode_piece(k);
            }
        };
        std::vector<std::thread> ppool;
        const int nt = (int)std::min<size_t>((size_t)nthreads, std::max<size_t>(nparts, 1));
        for (int t = 1; t < nt; ++t) ppool.emplace_back(pworker);
        pworker();
        for (auto &t : ppool) t.join();
    }
    lap("decode records");
    // ---- stitch
    {
        bool seen_unplaced = false, have_prev = false;
        int32_t prev_tid = -1, prev_pos = -1, prev_spos = -1;
        bam.rec_off.assign(nparts + 1, 0);
        bam.run_off.assign(nparts + 1, 0);
        for (size_t k = 0; k < nparts; ++k) {
            const Part &pt = bam.parts[k];
            int64_t at = INT64_MAX;
            std::string msg;
            if (pt.any_placed) {
                const int64_t f = pt.first_placed_rec;
                if (seen_unplaced || (have_prev && (pt.first_tid < prev_tid || (pt.first_tid == prev_tid && pt.first_pos < prev_pos)))) {
                    at = f; msg = "BAM file is not coordinate sorted: " + bam.path;
                } else if (have_prev && prev_tid == pt.first_tid && prev_spos > pt.first_spos && pt.err_rec != f) {
                    at = f; msg = "alignment starting with a deletion breaks coordinate order; not supported";
                }
            }
            if (pt.err_rec < at || (pt.err_rec == at && pt.err_before_order)) { at = pt.err_rec; msg = pt.err; }
            if (at != INT64_MAX) return fail(msg);
            if (pt.any_placed && !pt.tid.empty()) { have_prev = true; prev_tid = pt.last_tid; prev_pos = pt.last_pos; prev_spos = pt.last_spos; }
            seen_unplaced |= pt.saw_unplaced;
            bam.mapped += pt.mapped; bam.unplaced += pt.unplaced; bam.total += pt.total;
            bam.rec_off[k + 1] = bam.rec_off[k] + pt.tid.size();
            bam.run_off[k + 1] = bam.run_off[k] + pt.blk_start.size();
        }
        if (walk_truncated) return fail("truncated BAM record");
        bam.nrec = bam.rec_off[nparts];
        bam.nrun = bam.run_off[nparts];
        bam.threads = nthreads;
    }
    lap("stitch");
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
    counts[0] = (int64_t)b->nrec;
    counts[1] = (int64_t)b->nrun;
    counts[2] = b->mapped;
    counts[3] = b->total;
    return 0;
}

int pb_fill(void *h, int32_t *tid, int32_t *pos, uint16_t *alen, uint8_t *flags, uint8_t *nblk, int32_t *blk_start,
            int32_t *blk_len) {
    Bam *b = static_cast<Bam *>(h);
    if (!b || !b->loaded) return fail("pb_fill: file not loaded");
    const size_t nparts = b->parts.size();
    std::atomic<size_t> next(0);
    auto worker = [&]() {
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= nparts) return;
            const Part &pt = b->parts[k];
            const size_t at = b->rec_off[k], n = pt.tid.size(), rat = b->run_off[k], m = pt.blk_start.size();
            if (n) {
                std::memcpy(tid + at, pt.tid.data(), n * 4);
                std::memcpy(pos + at, pt.pos.data(), n * 4);
                std::memcpy(alen + at, pt.alen.data(), n * 2);
                std::memcpy(flags + at, pt.flags.data(), n);
                std::memcpy(nblk + at, pt.nblk.data(), n);
            }
            if (m) {
                std::memcpy(blk_start + rat, pt.blk_start.data(), m * 4);
                std::memcpy(blk_len + rat, pt.blk_len.data(), m * 4);
            }
        }
    };
    std::vector<std::thread> pool;
    const int nt = (int)std::min<size_t>((size_t)std::max(b->threads, 1), std::max<size_t>(nparts, 1));
    for (int t = 1; t < nt; ++t) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    return 0;
}

} // extern "C"
