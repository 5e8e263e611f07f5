// bam_stager.cpp -- native BAM -> packed-array stager (host side, no GPU code).
//
// The step *before* the counting path: the reference gets its reads from pysam/htslib
// (`pysam.AlignmentFile(X, "rb")`, `.fetch`, `read.positions`, `read.is_reverse`, `.mapped`;
// plastid/genomics/genome_array.py:660, 669, 690, 800-815).  pysam is not part of this
// product; this file reads a coordinate-sorted BAM (SAM/BAM spec v1: BGZF blocks, BAM
// records) straight into the flat arrays `pc_add_alignment_file` stages to HBM:
//   tid, pos, alen (= number of M/=/X reference positions), flags (bit0 = reverse strand),
//   nblk (maximal runs of contiguous aligned positions), runs of the gapped records.
// BGZF members are independent, so they are inflated (and the records in them decoded) by a pool of threads.
//
// C ABI (ctypes: plastid_amd/bam.py):
//   pb_open / pb_close, pb_nref / pb_ref_name / pb_ref_length,
//   pb_load  (decode the whole file), pb_counts, pb_fill (copy into caller arrays)
#include <dlfcn.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <chrono>
#include <mutex>
#include <thread>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
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
    // the columns: n records, nrun runs of the gapped ones, held by the load's Arena
    int32_t *tid = nullptr, *pos = nullptr, *blk_start = nullptr, *blk_len = nullptr;
    uint16_t *alen = nullptr;
    uint8_t *flags = nullptr, *nblk = nullptr;
    // what a read filter may look at beyond strand and length (genome_array.py:697-722: filters take the pysam read):
    // the SAM FLAG word, MAPQ and l_seq (pysam's query_length) of every staged record
    uint16_t *flag16 = nullptr;
    uint8_t *mapq = nullptr;
    int32_t *lseq = nullptr;
    // ... and the NH:i tag (number of reported alignments of the query; `read.get_tag("NH") == 1` is the usual unique-mapper
    // filter): its value clamped to 65 535, 0 when the record carries none
    uint16_t *nh = nullptr;
    size_t n = 0, nrun = 0;
    int64_t mapped = 0, unplaced = 0, total = 0;
    // wide records of the piece (more than 65 535 aligned positions or more than 255 aligned runs: markers 65535 / 255 in
    // the packed columns, true values here): record index within the piece, aligned length, run count
    std::vector<int64_t> wide_idx;
    std::vector<int32_t> wide_alen, wide_nblk;
    // first defect found inside the piece: record index within the piece, message; `before_order` marks
    // the checks the serial walk makes before it looks at the sort order of a record
    int64_t err_rec = INT64_MAX;
    bool err_before_order = false;
    std::string err;
    // what the stitching needs to know about the piece
    bool any_placed = false, saw_unplaced = false;
    int64_t first_placed_rec = -1;
    int32_t first_tid = -1, first_pos = -1, first_spos = -1;
    int32_t last_tid = -1, last_pos = -1, last_spos = -1;
    bool bad_size = false;            // decode_span stopped at a length prefix below the fixed record size
};

// Column storage of one load: 64 MiB anonymous mappings (on huge pages where the host grants them), handed
// out by bumping a pointer.  Thousands of per-piece vectors cost more in mmap/munmap and page faults than
// the decoding itself.  Regions released by a closed file wait in a process-wide pool (at most PB_POOL_MB,
// default 2048) for the next load -- several BAM files are the normal case -- which then neither faults
// them in again nor unmaps them.
struct RegionPool {
    static constexpr size_t kRegion = (size_t)64 << 20;
    std::mutex mu;
    std::vector<void *> idle;
    size_t cap = 2048;   // MiB
    RegionPool() { if (const char *env = getenv("PB_POOL_MB")) cap = (size_t)std::max(0, atoi(env)); }
    ~RegionPool() { for (void *q : idle) munmap(q, kRegion); }
    void *get() {
        {
            std::lock_guard<std::mutex> lock(mu);
            if (!idle.empty()) { void *q = idle.back(); idle.pop_back(); return q; }
        }
        void *q = mmap(nullptr, kRegion, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (q == MAP_FAILED) return nullptr;
        (void)madvise(q, kRegion, MADV_HUGEPAGE);
        return q;
    }
    void put(void *q) {
        {
            std::lock_guard<std::mutex> lock(mu);
            if ((idle.size() + 1) * (kRegion >> 20) <= cap) { idle.push_back(q); return; }
        }
        munmap(q, kRegion);
    }
};
RegionPool &region_pool() {
    static RegionPool pool;
    return pool;
}

struct Arena {
    std::mutex mu;
    std::vector<std::pair<void *, size_t>> regions;
    uint8_t *cur = nullptr;
    size_t left = 0;
    void *alloc(size_t bytes) {
        bytes = (bytes + 63) & ~(size_t)63;
        std::lock_guard<std::mutex> lock(mu);
        if (bytes > left) {
            void *q;
            size_t sz = RegionPool::kRegion;
            if (bytes <= sz) {
                q = region_pool().get();
            } else {                                      // an outsize piece gets a mapping of its own
                sz = (bytes + 4095) & ~(size_t)4095;
                q = mmap(nullptr, sz, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
                if (q == MAP_FAILED) q = nullptr;
            }
            if (!q) return nullptr;
            regions.emplace_back(q, sz);
            cur = (uint8_t *)q;
            left = sz;
        }
        void *r = cur;
        cur += bytes;
        left -= bytes;
        return r;
    }
    void clear() {
        for (auto &r : regions) {
            if (r.second == RegionPool::kRegion) region_pool().put(r.first);
            else munmap(r.first, r.second);
        }
        regions.clear();
        cur = nullptr;
        left = 0;
    }
    ~Arena() { clear(); }
};

// The records of the piece being decoded (scratch: reused from piece to piece, so it stays in cache; the
// finished columns are copied into the Arena at their exact size).
struct Cols {
    std::vector<int32_t> tid, pos, blk_start, blk_len;
    std::vector<uint16_t> alen, flag16, nh;
    std::vector<uint8_t> flags, nblk, mapq;
    std::vector<int32_t> lseq;
    size_t n = 0;   // records held: the five per-record columns are sized for the piece up front and written by index
    void clear() { n = 0; blk_start.clear(); blk_len.clear(); }
    void room(size_t records) {
        if (tid.size() < records) { tid.resize(records); pos.resize(records); alen.resize(records); flags.resize(records); nblk.resize(records);
                                      flag16.resize(records); mapq.resize(records); lseq.resize(records); nh.resize(records); }
    }
};

struct Bam {
    std::string path;
    Arena arena;
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
// a read-only view of a whole file (mapped: the inflate threads read the page cache directly)
struct FileMap {
    const uint8_t *p = nullptr;
    size_t n = 0;
    bool ok = false;
    explicit FileMap(const std::string &path) {
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) return;
        struct stat st;
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
            n = (size_t)st.st_size;
            if (n == 0) ok = true;
            else {
                void *q = mmap(nullptr, n, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
                if (q != MAP_FAILED) { p = (const uint8_t *)q; ok = true; }
            }
        }
        close(fd);
    }
    ~FileMap() { if (p) munmap(const_cast<uint8_t *>(p), n); }
    const uint8_t *data() const { return p; }
    size_t size() const { return n; }
};

int scan_blocks(const FileMap &file, std::vector<Block> &blocks, size_t &total_u) {
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
            if (sf[0] == 'B' && sf[1] == 'C' && slen == 2 && x + 6 <= xlen) bsize = rd16(sf + 4);   // (the two payload bytes lie inside the extra field)
            x += 4 + slen;
        }
        if (bsize < 0) return fail("BGZF member without BC subfield");
        const size_t clen = (size_t)bsize + 1;
        if (off + clen > file.size()) return fail("truncated BGZF member");
        const uint32_t isize = rd32(file.data() + off + clen - 4);
        if (isize > (1u << 16)) return fail("corrupt BGZF member (more than 64 KiB of payload)");
        blocks.push_back({off, (uint32_t)clen, isize, total_u});
        total_u += isize;
        off += clen;
    }
    return 0;
}

// libdeflate, when the host has it (looked up at run time; nothing links against it): its raw-deflate
// decoder and its carry-less-multiply CRC-32 are several times faster than zlib 1.2's.  PB_ZLIB=1 keeps
// to zlib.  One decompressor per thread.
struct LibDeflate {
    void *(*alloc)() = nullptr;
    int (*decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    void (*release)(void *) = nullptr;
    uint32_t (*crc32)(uint32_t, const void *, size_t) = nullptr;
    bool ok = false;
    LibDeflate() {
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = reinterpret_cast<decltype(alloc)>(dlsym(h, "libdeflate_alloc_decompressor"));
        decompress = reinterpret_cast<decltype(decompress)>(dlsym(h, "libdeflate_deflate_decompress"));
        release = reinterpret_cast<decltype(release)>(dlsym(h, "libdeflate_free_decompressor"));
        crc32 = reinterpret_cast<decltype(crc32)>(dlsym(h, "libdeflate_crc32"));
        ok = alloc && decompress && release && crc32;
    }
};
const LibDeflate &libdeflate() {
    static const LibDeflate ld;
    return ld;
}
struct ThreadDecompressor {
    void *d = nullptr;
    ~ThreadDecompressor() { if (d) libdeflate().release(d); }
};

// raw deflate stream -> exactly `out_n` bytes with CRC-32 `crc`; 0, -1 (damaged stream) or -2 (CRC)
int raw_inflate(const uint8_t *in, size_t in_n, uint8_t *out, size_t out_n, uint32_t crc) {
    const LibDeflate &ld = libdeflate();
    if (ld.ok && getenv("PB_ZLIB") == nullptr) {
        thread_local ThreadDecompressor td;
        if (!td.d) td.d = ld.alloc();
        if (td.d) {
            if (ld.decompress(td.d, in, in_n, out, out_n, nullptr) != 0) return -1;
            return ld.crc32(0, out, out_n) == crc ? 0 : -2;
        }
    }
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) return -1;
    zs.next_in = const_cast<uint8_t *>(in);
    zs.avail_in = (uInt)in_n;
    zs.next_out = out;
    zs.avail_out = (uInt)out_n;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    if (rc != Z_STREAM_END || zs.avail_out != 0) return -1;
    return (uint32_t)::crc32(::crc32(0L, Z_NULL, 0), out, (uInt)out_n) == crc ? 0 : -2;
}

int inflate_block(const FileMap &file, const Block &b, uint8_t *dst) {
    if (b.ulen == 0) return 0;
    const uint8_t *h = file.data() + b.coff;
    const size_t hdr = 12 + rd16(h + 10);
    if (b.clen < hdr + 8) return -1;
    return raw_inflate(h + hdr, b.clen - hdr - 8, dst, b.ulen, rd32(h + b.clen - 8));
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

// BAM header (magic, text, reference list) at `p`; leaves `p` at the first alignment record.
// need_more (optional): set when the buffer ends inside the header (a caller that inflates lazily
// fetches more and retries) instead of reporting truncation.
int parse_header(Bam &bam, const uint8_t *&p, const uint8_t *end, uint32_t &n_ref_out, bool *need_more = nullptr) {
    auto trunc = [&](const char *what) { if (need_more) { *need_more = true; return -1; } return fail(what); };
    if (end - p < 12) return trunc("not a BAM file (bad magic)");
    if (std::memcmp(p, "BAM\1", 4) != 0) return fail("not a BAM file (bad magic)");
    const uint32_t l_text = rd32(p + 4);
    p += 8;
    if ((size_t)(end - p) < (size_t)l_text + 4) return trunc("truncated BAM header");
    p += l_text;
    const uint32_t n_ref = rd32(p);
    p += 4;
    bam.ref_names.clear();
    bam.ref_lengths.clear();
    for (uint32_t r = 0; r < n_ref; ++r) {
        if (end - p < 4) return trunc("truncated BAM reference list");
        const uint32_t l_name = rd32(p);
        p += 4;
        if ((size_t)(end - p) < (size_t)l_name + 4) return trunc("truncated BAM reference list");
        bam.ref_names.emplace_back(reinterpret_cast<const char *>(p), l_name ? l_name - 1 : 0);
        p += l_name;
        bam.ref_lengths.push_back((int32_t)rd32(p));
        p += 4;
    }
    n_ref_out = n_ref;
    return 0;
}

// Decodes the records that start at `q` and lie wholly inside [q, limit), at most `max_rec` of them, into
// `pt` with every per-record format check and the order checks inside the piece.  Returns where it
// stopped: `limit`, the start of the first record that crosses `limit`, or the record whose length
// prefix is impossible (pt.bad_size).  A defect ends the piece (pt.err, pt.err_rec).
// NH:i of a record: the auxiliary fields follow the packed sequence ((l_seq + 1) / 2 bytes) and the qualities (l_seq bytes)
// behind the CIGAR; each is {tag[2], type, value} (SAM spec 4.2.4; walked as htslib's bam_aux_get does, sam.c:915-965 in the
// vendored tree: A c C one byte, s S two, i I f four, Z H NUL-terminated, B {subtype, count:u32, elements}).  The value of
// an integer-typed NH, clamped to [0, 65 535]; 0 when the record has no such field (or the fields are malformed).
static inline uint16_t aux_nh(const uint8_t *after_cigar, int32_t l_seq, const uint8_t *end) {
    if (l_seq < 0) return 0;
    const uint8_t *a = after_cigar + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
    while (a + 3 <= end && a >= after_cigar) {
        const uint8_t t0 = a[0], t1 = a[1], type = a[2];
        a += 3;
        size_t sz;
        switch (type) {
        case 'A': case 'c': case 'C': sz = 1; break;
        case 's': case 'S': sz = 2; break;
        case 'i': case 'I': case 'f': sz = 4; break;
        case 'Z': case 'H': { const uint8_t *z = a; while (z < end && *z) ++z; if (z >= end) return 0; sz = (size_t)(z - a) + 1; break; }
        case 'B': {
            if (a + 5 > end) return 0;
            const uint8_t sub = a[0];
            const uint32_t cnt = rd32(a + 1);
            const size_t es = (sub == 'c' || sub == 'C') ? 1 : ((sub == 's' || sub == 'S') ? 2 : ((sub == 'i' || sub == 'I' || sub == 'f') ? 4 : 0));
            if (!es) return 0;
            sz = 5 + (size_t)cnt * es;
            break;
        }
        default: return 0;
        }
        if ((size_t)(end - a) < sz) return 0;
        if (t0 == 'N' && t1 == 'H') {
            int64_t v;
            switch (type) {
            case 'c': v = (int8_t)a[0]; break;
            case 'C': v = a[0]; break;
            case 's': v = (int16_t)rd16(a); break;
            case 'S': v = rd16(a); break;
            case 'i': v = (int32_t)rd32(a); break;
            case 'I': v = rd32(a); break;
            default: return 0;
            }
            return (uint16_t)(v < 0 ? 0 : (v > 65535 ? 65535 : v));
        }
        a += sz;
    }
    return 0;
}

const uint8_t *decode_span_cols(const Bam &bam, Part &pt, Cols &cols, const uint8_t *q, const uint8_t *limit, uint32_t n_ref, int64_t max_rec) {
    std::vector<std::pair<int32_t, int32_t>> runs;
    cols.clear();
    cols.room(limit > q ? (size_t)(limit - q) / 36 + 1 : 1);   // a record takes at least 36 bytes
    // the state every record touches lives in locals (the columns are int32 / uint8 arrays: through `pt` and `cols`
    // the compiler has to assume that every store may change them) and is written back on the way out
    int32_t *const c_tid = cols.tid.data(), *const c_pos = cols.pos.data();
    uint16_t *const c_alen = cols.alen.data();
    uint8_t *const c_flags = cols.flags.data(), *const c_nblk = cols.nblk.data(), *const c_mapq = cols.mapq.data();
    uint16_t *const c_flag16 = cols.flag16.data();
    int32_t *const c_lseq = cols.lseq.data();
    uint16_t *const c_nh = cols.nh.data();
    size_t n = 0;
    int64_t total = pt.total, mapped = pt.mapped, unplaced = pt.unplaced;
    bool any_placed = pt.any_placed, saw_unplaced = pt.saw_unplaced;
    int32_t last_tid = pt.last_tid, last_pos = pt.last_pos, last_spos = pt.last_spos;
    const uint8_t *ret = nullptr;
    auto bad = [&](int64_t i, bool before_order, const std::string &m) {
        pt.err_rec = i; pt.err_before_order = before_order; pt.err = m;
        return q;
    };
    for (int64_t i = 0; i < max_rec; ++i) {
        if (limit - q < 4) break;
        const uint32_t block_size = rd32(q);
        if (block_size < 32) { pt.bad_size = true; break; }
        if ((size_t)(limit - q - 4) < block_size) break;
        const uint8_t *r = q + 4;
        q = r + block_size;
        const int32_t tid = (int32_t)rd32(r), pos = (int32_t)rd32(r + 4);
        const uint8_t l_read_name = r[8], mapq = r[9];
        const uint16_t n_cigar = rd16(r + 12), flag = rd16(r + 14);
        const int32_t l_seq = (int32_t)rd32(r + 16);
        total += 1;
        if (!(flag & 0x4)) mapped += 1;
        if (tid < 0) { // unplaced reads sit at the end of a sorted BAM; fetch() never returns them
            unplaced += 1;
            saw_unplaced = true;
            continue;
        }
        if (tid >= (int32_t)n_ref) { ret = bad(i, true, "BAM record with reference id out of range"); break; }
        if (pos < 0) { ret = bad(i, true, "placed BAM record with a negative position"); break; }
        if (!any_placed) {        // its order against the previous piece is checked when stitching
            any_placed = true;
            pt.first_placed_rec = i;
            pt.first_tid = tid; pt.first_pos = pos;
            if (saw_unplaced) { ret = bad(i, false, "BAM file is not coordinate sorted: " + bam.path); break; }
        } else if (saw_unplaced || tid < last_tid || (tid == last_tid && pos < last_pos)) {
            ret = bad(i, false, "BAM file is not coordinate sorted: " + bam.path);
            break;
        }
        const bool first = n == 0;
        if ((size_t)32 + l_read_name + (size_t)n_cigar * 4 > block_size) { ret = bad(i, false, "corrupt BAM record (cigar overruns block)"); break; }
        const uint8_t *cig = r + 32 + l_read_name;
        const uint16_t nh = aux_nh(cig + (size_t)n_cigar * 4, l_seq, q);
        if (n_cigar == 1) {   // the common record: one M / = / X operation (one aligned run starting at pos)
            const uint32_t v = rd32(cig), op = v & 0xf, len = v >> 4;
            if ((op == 0 || op == 7 || op == 8) && len > 0 && len <= 65535 && (int64_t)pos + len <= 0x7fffffffLL) {
                if (first) pt.first_spos = pos;
                else if (last_tid == tid && last_spos > pos) {
                    ret = bad(i, false, "alignment starting with a deletion breaks coordinate order; not supported");
                    break;
                }
                last_tid = tid; last_pos = pos; last_spos = pos;
                c_tid[n] = tid;
                c_pos[n] = pos;
                c_alen[n] = (uint16_t)len;
                c_flags[n] = (flag & 0x10) ? 1 : 0;
                c_nblk[n] = 1;
                c_flag16[n] = flag; c_mapq[n] = mapq; c_lseq[n] = l_seq; c_nh[n] = nh;
                ++n;
                continue;
            }
        }
        runs.clear();
        int64_t ref = pos, L = 0;
        bool unknown_op = false;
        for (uint16_t c = 0; c < n_cigar && !unknown_op; ++c) {
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
                unknown_op = true;
                break;
            }
        }
        if (unknown_op) { ret = bad(i, false, "unknown CIGAR operation in " + bam.path); break; }
        if (ref > 0x7fffffffLL) { ret = bad(i, false, "alignment ends beyond 2^31 - 1"); break; }   // run starts are int32
        if (L > 0x7fffffffLL) { ret = bad(i, false, "alignment with more than 2^31 - 1 aligned positions"); break; }
        // beyond the 16-bit / 8-bit columns (or equal to both markers): a wide record, its true values aside
        const bool wide = L > 65535 || runs.size() > 255 || (L == 65535 && runs.size() == 255);
        if (wide) { pt.wide_idx.push_back((int64_t)n); pt.wide_alen.push_back((int32_t)L); pt.wide_nblk.push_back((int32_t)runs.size()); }
        // the packed format keys a record on its first aligned position; a CIGAR that opens with
        // D/N (not produced by aligners) is accepted only if that keeps the file order
        const int32_t spos = runs.empty() ? pos : runs[0].first;
        if (first) pt.first_spos = spos;
        else if (last_tid == tid && last_spos > spos) {
            ret = bad(i, false, "alignment starting with a deletion breaks coordinate order; not supported");
            break;
        }
        last_tid = tid; last_pos = pos; last_spos = spos;
        c_tid[n] = tid;
        c_pos[n] = spos;
        c_alen[n] = wide ? (uint16_t)65535 : (uint16_t)L;
        c_flags[n] = (flag & 0x10) ? 1 : 0;
        c_nblk[n] = wide ? (uint8_t)255 : (uint8_t)runs.size();
        c_flag16[n] = flag; c_mapq[n] = mapq; c_lseq[n] = l_seq; c_nh[n] = nh;
        ++n;
        if (runs.size() >= 2)
            for (auto &x : runs) {
                cols.blk_start.push_back(x.first);
                cols.blk_len.push_back(x.second);
            }
    }
    cols.n = n;
    pt.total = total; pt.mapped = mapped; pt.unplaced = unplaced;
    pt.any_placed = any_placed; pt.saw_unplaced = saw_unplaced;
    pt.last_tid = last_tid; pt.last_pos = last_pos; pt.last_spos = last_spos;
    return ret ? ret : q;
}

const uint8_t *decode_span(Bam &bam, Part &pt, const uint8_t *q, const uint8_t *limit, uint32_t n_ref, int64_t max_rec) {
    thread_local Cols cols;
    const uint8_t *stop = decode_span_cols(bam, pt, cols, q, limit, n_ref, max_rec);
    // the finished columns, at their exact size, into the load's arena
    const size_t n = cols.n, m = cols.blk_start.size();
    if (n) {
        uint8_t *mem = (uint8_t *)bam.arena.alloc(n * 21 + 64 * 9 + m * 8 + 64 * 2);
        if (!mem) {
            if (pt.err_rec == INT64_MAX) { pt.err_rec = 0; pt.err_before_order = true; pt.err = "out of memory reading " + bam.path; }
            return stop;
        }
        auto take = [&](size_t bytes) { uint8_t *r = mem; mem += (bytes + 63) & ~(size_t)63; return r; };
        pt.tid = (int32_t *)take(n * 4); pt.pos = (int32_t *)take(n * 4); pt.alen = (uint16_t *)take(n * 2);
        pt.flags = take(n); pt.nblk = take(n);
        std::memcpy(pt.tid, cols.tid.data(), n * 4); std::memcpy(pt.pos, cols.pos.data(), n * 4);
        std::memcpy(pt.alen, cols.alen.data(), n * 2); std::memcpy(pt.flags, cols.flags.data(), n); std::memcpy(pt.nblk, cols.nblk.data(), n);
        pt.flag16 = (uint16_t *)take(n * 2); pt.mapq = take(n); pt.lseq = (int32_t *)take(n * 4);
        std::memcpy(pt.flag16, cols.flag16.data(), n * 2); std::memcpy(pt.mapq, cols.mapq.data(), n); std::memcpy(pt.lseq, cols.lseq.data(), n * 4);
        pt.nh = (uint16_t *)take(n * 2);
        std::memcpy(pt.nh, cols.nh.data(), n * 2);
        if (m) {
            pt.blk_start = (int32_t *)take(m * 4); pt.blk_len = (int32_t *)take(m * 4);
            std::memcpy(pt.blk_start, cols.blk_start.data(), m * 4); std::memcpy(pt.blk_len, cols.blk_len.data(), m * 4);
        }
    }
    pt.n = n;
    pt.nrun = m;
    return stop;
}

// The checks that span two pieces, and the flat offsets of every piece (bam.parts in file order).  The
// defect reported is the one of the lowest record index, as in a serial walk (where one record fails
// two checks, the serial order of the checks decides).
int stitch_parts(Bam &bam, bool truncated, int nthreads) {
    const size_t nparts = bam.parts.size();
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
        if (pt.any_placed && pt.n > 0) { have_prev = true; prev_tid = pt.last_tid; prev_pos = pt.last_pos; prev_spos = pt.last_spos; }
        seen_unplaced |= pt.saw_unplaced;
        bam.mapped += pt.mapped; bam.unplaced += pt.unplaced; bam.total += pt.total;
        bam.rec_off[k + 1] = bam.rec_off[k] + pt.n;
        bam.run_off[k + 1] = bam.run_off[k] + pt.nrun;
    }
    if (truncated) return fail("truncated BAM record");
    bam.nrec = bam.rec_off[nparts];
    bam.nrun = bam.run_off[nparts];
    bam.threads = nthreads;
    return 0;
}

// Alignment records in [p, end) -> bam.parts (file order), with every order / format check.
int decode_records(Bam &bam, const uint8_t *p, const uint8_t *end, uint32_t n_ref, int nthreads, Lap &lap) {
    // A serial walk over the length prefixes cuts the stream into pieces of kPiece records; the pieces
    // are decoded by the thread pool and stitched.  (The region loader's path: its buffers are small.
    // Whole files go through decode(), which needs no serial walk.)
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
    {
        std::atomic<size_t> nextp(0);
        auto pworker = [&]() {
            for (;;) {
                const size_t k = nextp.fetch_add(1);
                if (k >= nparts) return;
                decode_span(bam, bam.parts[k], cuts[k], cuts[k + 1], n_ref, INT64_MAX);
            }
        };
        std::vector<std::thread> ppool;
        const int nt = (int)std::min<size_t>((size_t)nthreads, std::max<size_t>(nparts, 1));
        for (int t = 1; t < nt; ++t) ppool.emplace_back(pworker);
        pworker();
        for (auto &t : ppool) t.join();
    }
    lap("decode records");
    const int rc = stitch_parts(bam, walk_truncated, nthreads);
    lap("stitch");
    return rc;
}

// Where a record starts in [lo, hi), for a worker that holds a chunk of the inflated stream but not
// what precedes it: the first offset from which three records in a row look like BAM records (or
// fewer, when they reach the end of the chunk).  Only a guess -- decode() keeps what was decoded from
// it only if the preceding chunk's last record ends exactly there.
const uint8_t *guess_record_start(const uint8_t *lo, const uint8_t *hi, uint32_t n_ref) {
    auto plausible = [&](const uint8_t *q, const uint8_t *&next) {
        if (hi - q < 36) return false;
        const uint32_t bs = rd32(q);
        if (bs < 32 || bs > (1u << 26)) return false;
        const int32_t tid = (int32_t)rd32(q + 4), pos = (int32_t)rd32(q + 8);
        const uint32_t l_name = q[12], n_cigar = rd16(q + 16);
        const int32_t l_seq = (int32_t)rd32(q + 20), mtid = (int32_t)rd32(q + 24), mpos = (int32_t)rd32(q + 28);
        if (tid < -1 || tid >= (int32_t)n_ref || mtid < -1 || mtid >= (int32_t)n_ref || pos < -1 || mpos < -1) return false;
        if (l_name == 0 || l_seq < 0 || l_seq > (1 << 28)) return false;
        if ((uint64_t)32 + l_name + 4ull * n_cigar + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq > bs) return false;
        const uint8_t *nul = q + 36 + l_name - 1;
        if (nul < hi && *nul != 0) return false;
        next = q + 4 + (size_t)bs;
        return true;
    };
    const uint8_t *stop = hi - lo > (1 << 20) ? lo + (1 << 20) : hi;
    for (const uint8_t *q = lo; q < stop; ++q) {
        const uint8_t *a = q, *next = nullptr;
        int k = 0;
        while (k < 3 && plausible(a, next)) {
            ++k;
            if (next >= hi) { k = 3; break; }
            a = next;
        }
        if (k == 3) return q;
    }
    return nullptr;
}

// Whole file.  The BGZF members are grouped into chunks of about 1 MiB of inflated stream; a worker
// inflates a chunk into its own scratch buffer and, while it is still in its cache, decodes the
// records that start in it from a guessed record boundary (guess_record_start).  The inflated stream
// is never held as a whole: of each chunk only the first bytes and the bytes behind its last whole
// record are kept.  The serial pass that follows only chains the chunks: it confirms each guess
// against the end of the preceding chunk's last record (inflating and decoding the chunk again from
// the right offset where a guess was wrong), and decodes the one record that straddles each chunk
// boundary.  No pass walks the whole stream on one thread.  Positions are offsets in the inflated
// stream.
int decode(Bam &bam, int nthreads) {
    Lap lap;
    FileMap file(bam.path);
    if (!file.ok) return fail("cannot read " + bam.path);
    std::vector<Block> blocks;
    size_t total_u = 0;
    if (scan_blocks(file, blocks, total_u) != 0) return -1;
    lap("map + index members");
    auto inflate_error = [&](int rc) { return fail(std::string(rc == -2 ? "BGZF CRC mismatch in " : "BGZF inflate failed in ") + bam.path); };

    // ---- BAM header: inflate members from the start until it is complete
    uint32_t n_ref = 0;
    size_t pre = 0;                       // members inflated here
    std::vector<uint8_t> hbuf;
    size_t first_rec = 0;
    for (;;) {
        if (pre < blocks.size()) {
            hbuf.resize(blocks[pre].uoff + blocks[pre].ulen);
            const int rc = inflate_block(file, blocks[pre], hbuf.data() + blocks[pre].uoff);
            if (rc != 0) return inflate_error(rc);
            ++pre;
        }
        const uint8_t *p = hbuf.data();
        bool more = false;
        if (parse_header(bam, p, hbuf.data() + hbuf.size(), n_ref, pre < blocks.size() ? &more : nullptr) == 0) {
            first_rec = (size_t)(p - hbuf.data());
            break;
        }
        if (!more) return -1;
    }
    lap("header");

    // ---- chunks of members; chunk 0 = what the header pass inflated (its first record is known)
    size_t chunk_bytes = (size_t)1 << 20;
    if (const char *env = getenv("PB_CHUNK")) chunk_bytes = (size_t)std::max(1, atoi(env));   // test knob: tiny chunks exercise the chaining
    const size_t kNone = (size_t)-1;
    size_t kHead = (size_t)32 << 10;
    if (const char *env = getenv("PB_HEAD")) kHead = (size_t)std::max(0, atoi(env));            // test knob: records longer than a head
    struct Chunk {
        size_t b0, b1;                    // members
        size_t lo, hi;                    // its bytes of the inflated stream
        size_t guess, stop;               // where decoding started / stopped
        std::vector<uint8_t> head, tail;  // stream bytes [lo, lo + head.size()) and [stop, hi)
        Part part;
    };
    std::vector<Chunk> chunks;
    size_t max_chunk = 0;
    {
        Chunk c0;
        c0.b0 = 0; c0.b1 = pre; c0.lo = 0; c0.hi = hbuf.size();
        c0.guess = c0.stop = kNone;
        chunks.push_back(std::move(c0));
        for (size_t b = pre; b < blocks.size();) {
            Chunk c;
            c.b0 = b;
            size_t bytes = 0;
            while (b < blocks.size() && (bytes < chunk_bytes || blocks[b].ulen == 0)) bytes += blocks[b++].ulen;
            c.b1 = b;
            c.lo = blocks[c.b0].uoff;
            c.hi = c.lo + bytes;
            c.guess = c.stop = kNone;
            max_chunk = std::max(max_chunk, bytes);
            chunks.push_back(std::move(c));
        }
    }
    // chunk k inflated into buf (resized to its length)
    auto inflate_chunk = [&](const Chunk &c, std::vector<uint8_t> &buf) {
        buf.resize(c.hi - c.lo);
        for (size_t b = c.b0; b < c.b1; ++b) {
            const int rc = inflate_block(file, blocks[b], buf.data() + (blocks[b].uoff - c.lo));
            if (rc != 0) return rc;
        }
        return 0;
    };
    // decode chunk c, held in `base` (= stream offset c.lo), from stream offset `from`; keeps head and tail
    auto decode_chunk = [&](Chunk &c, const uint8_t *base, size_t from) {
        const size_t len = c.hi - c.lo;
        c.part = Part();
        const uint8_t *q = decode_span(bam, c.part, base + (from - c.lo), base + len, n_ref, INT64_MAX);
        c.guess = from;
        c.stop = c.lo + (size_t)(q - base);
        c.head.assign(base, base + std::min(len, kHead));
        c.tail.assign(q, base + len);
    };
    if (first_rec < chunks[0].hi) decode_chunk(chunks[0], hbuf.data(), first_rec);
    std::atomic<size_t> next(1);
    std::atomic<int> bad(0);
    if (nthreads < 1) nthreads = 1;
    nthreads = (int)std::min<size_t>((size_t)nthreads, chunks.size());
    const bool timing = getenv("PB_TIMING") != nullptr;
    std::atomic<int64_t> ns_inflate(0), ns_decode(0), ns_start(0);
    const auto t_spawn = std::chrono::steady_clock::now();
    auto worker = [&]() {
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto t0 = now();
        if (timing) ns_start += std::chrono::duration_cast<std::chrono::nanoseconds>(t0 - t_spawn).count();
        std::vector<uint8_t> buf;
        buf.reserve(max_chunk);
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= chunks.size()) return;
            Chunk &c = chunks[k];
            if (timing) t0 = now();
            const int rc = inflate_chunk(c, buf);
            if (rc != 0) { bad.store(rc); continue; }
            if (bad.load() != 0) continue;
            auto t1 = t0;
            if (timing) { t1 = now(); ns_inflate += std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count(); }
            const uint8_t *g = guess_record_start(buf.data(), buf.data() + buf.size(), n_ref);
            if (g) decode_chunk(c, buf.data(), c.lo + (size_t)(g - buf.data()));
            else c.head.assign(buf.data(), buf.data() + std::min(buf.size(), kHead));
            if (timing) ns_decode += std::chrono::duration_cast<std::chrono::nanoseconds>(now() - t1).count();
        }
    };
    {
        std::vector<std::thread> pool;
        for (int t = 1; t < nthreads; ++t) pool.emplace_back(worker);
        worker();
        for (auto &t : pool) t.join();
    }
    if (bad.load() != 0) return inflate_error(bad.load());
    if (timing)
        fprintf(stderr, "[bam] %d threads: summed over threads, inflate %.1f ms, decode %.1f ms; mean start delay %.2f ms\n", nthreads,
                ns_inflate.load() / 1e6, ns_decode.load() / 1e6, ns_start.load() / 1e6 / nthreads);
    lap("inflate + decode chunks");

    // ---- chain the chunks
    bam.parts.clear();
    bam.parts.reserve(2 * chunks.size());
    bool truncated = false;
    int64_t redone = 0, refetched = 0;
    size_t expected = first_rec;
    std::vector<uint8_t> buf, rec;
    size_t buf_holds = kNone;                                // the chunk `buf` holds inflated
    for (size_t k = 0; k < chunks.size(); ++k) {
        Chunk &c = chunks[k];
        if (expected >= c.hi) continue;                      // a record (or the header) covers the whole chunk
        if (c.guess != expected) {                           // decode it from where the records really resume
            if (buf_holds != k) {
                if (k == 0) buf = hbuf;
                else if (const int rc = inflate_chunk(c, buf)) return inflate_error(rc);
                buf_holds = k;
            }
            decode_chunk(c, buf.data(), expected);
            ++redone;
        }
        const bool defect = c.part.err_rec != INT64_MAX, bad_size = c.part.bad_size;
        const size_t q = c.stop;
        bam.parts.push_back(std::move(c.part));
        if (defect) break;                                   // stitch_parts reports it
        if (bad_size) { truncated = true; break; }
        expected = q;
        if (q == c.hi) continue;
        // the record that starts at q and ends in a later chunk: its first bytes are c.tail, the rest
        // comes from the heads of the following chunks (or, for a record longer than a head, from
        // inflating them again)
        rec = c.tail;
        size_t j = k + 1, used = 0;                          // next unread stream byte: chunk j, offset `used`
        int pull_rc = 0;
        auto pull = [&](size_t want) {                       // appends up to `want` further stream bytes to rec
            while (want > 0 && j < chunks.size()) {
                const Chunk &d = chunks[j];
                const size_t len = d.hi - d.lo;
                if (used >= len) { ++j; used = 0; continue; }
                const size_t take = std::min(want, len - used);
                if (used + take <= d.head.size()) {
                    rec.insert(rec.end(), d.head.begin() + (long)used, d.head.begin() + (long)(used + take));
                } else {
                    if (buf_holds != j) {
                        if (j == 0) buf = hbuf;
                        else if ((pull_rc = inflate_chunk(d, buf)) != 0) return;
                        buf_holds = j;
                        ++refetched;
                    }
                    rec.insert(rec.end(), buf.begin() + (long)used, buf.begin() + (long)(used + take));
                }
                used += take;
                want -= take;
            }
        };
        if (rec.size() < 4) pull(4 - rec.size());
        if (pull_rc != 0) return inflate_error(pull_rc);
        if (rec.size() < 4) { truncated = true; break; }
        const uint32_t block_size = rd32(rec.data());
        const size_t need = 4 + (size_t)block_size;
        if (block_size < 32 || need > total_u - q) { truncated = true; break; }
        if (rec.size() < need) pull(need - rec.size());
        if (pull_rc != 0) return inflate_error(pull_rc);
        if (rec.size() < need) { truncated = true; break; }
        Part tail;
        decode_span(bam, tail, rec.data(), rec.data() + need, n_ref, 1);
        const bool tail_defect = tail.err_rec != INT64_MAX;
        bam.parts.push_back(std::move(tail));
        if (tail_defect) break;
        expected = q + need;
    }
    if (getenv("PB_TIMING"))
        fprintf(stderr, "[bam] %zu chunks, %lld decoded again serially, %lld inflated again for a long record\n", chunks.size(),
                (long long)redone, (long long)refetched);
    lap("chain chunks");
    if (stitch_parts(bam, truncated, nthreads) != 0) return -1;
    lap("stitch");
    bam.loaded = true;
    return 0;
}

// ---------------------------------------------------------------- region-limited loading (BAI)
// The reference fetches per region through the BAM index (genome_array.py:800-809 via pysam; bin /
// chunk / linear-index scheme of the SAM specification, section 5).  Here the regions of a whole
// query set are resolved at once: bins -> chunks of virtual offsets, clipped by the linear index,
// merged; only the BGZF members those chunks touch are read and inflated; the records then go
// through the same decoder as a whole file and are filtered by htslib's overlap test
// (pos < end && endpos > start).  `mapped` comes from the index's per-reference counts, as
// pysam's AlignmentFile.mapped does.
struct BaiRef {
    std::vector<uint32_t> bin_id;
    std::vector<std::vector<std::pair<uint64_t, uint64_t>>> chunks;
    std::vector<uint64_t> linear;
    uint64_t n_mapped = 0, n_unmapped = 0;
    bool has_meta = false;
};

int load_bai(const std::string &bam_path, std::vector<BaiRef> &refs) {
    std::vector<uint8_t> buf;
    std::string ipath = bam_path + ".bai";
    if (!read_file(ipath, buf)) {
        ipath = bam_path.size() > 4 ? bam_path.substr(0, bam_path.size() - 4) + ".bai" : ipath;
        if (!read_file(ipath, buf)) return fail("cannot read the index of " + bam_path + " (.bam.bai / .bai)");
    }
    const uint8_t *p = buf.data(), *end = p + buf.size();
    auto need = [&](size_t n) { return (size_t)(end - p) >= n; };
    if (!need(8) || std::memcmp(p, "BAI\1", 4) != 0) return fail("not a BAI index: " + ipath);
    const uint32_t n_ref = rd32(p + 4);
    p += 8;
    if (n_ref > (1u << 24)) return fail("corrupt BAI index (reference count): " + ipath);
    refs.assign(n_ref, BaiRef());
    auto rd64 = [](const uint8_t *q) { return (uint64_t)rd32(q) | ((uint64_t)rd32(q + 4) << 32); };
    for (uint32_t r = 0; r < n_ref; ++r) {
        BaiRef &br = refs[r];
        if (!need(4)) return fail("truncated BAI index: " + ipath);
        const uint32_t n_bin = rd32(p);
        p += 4;
        for (uint32_t b = 0; b < n_bin; ++b) {
            if (!need(8)) return fail("truncated BAI index: " + ipath);
            const uint32_t bin = rd32(p), n_chunk = rd32(p + 4);
            p += 8;
            if (!need((size_t)n_chunk * 16) || n_chunk > (1u << 28)) return fail("truncated BAI index: " + ipath);
            if (bin == 37450u) { // samtools' metadata pseudo-bin: (file range), (mapped, unmapped)
                if (n_chunk >= 2) { br.n_mapped = rd64(p + 16); br.n_unmapped = rd64(p + 24); br.has_meta = true; }
            } else {
                br.bin_id.push_back(bin);
                br.chunks.emplace_back();
                auto &v = br.chunks.back();
                v.reserve(n_chunk);
                for (uint32_t c = 0; c < n_chunk; ++c) v.emplace_back(rd64(p + 16 * (size_t)c), rd64(p + 16 * (size_t)c + 8));
            }
            p += (size_t)n_chunk * 16;
        }
        if (!need(4)) return fail("truncated BAI index: " + ipath);
        const uint32_t n_intv = rd32(p);
        p += 4;
        if (!need((size_t)n_intv * 8) || n_intv > (1u << 28)) return fail("truncated BAI index: " + ipath);
        br.linear.resize(n_intv);
        for (uint32_t i = 0; i < n_intv; ++i) br.linear[i] = rd64(p + 8 * (size_t)i);
        p += (size_t)n_intv * 8;
    }
    return 0;
}

// the BGZF member at file offset `coff`, inflated and appended to `out`; returns its compressed
// length, 0 at end of file, -1 on a damaged member
long read_member(FILE *f, uint64_t coff, std::vector<uint8_t> &out) {
    uint8_t h[18];
    if (fseeko(f, (off_t)coff, SEEK_SET) != 0) return -1;
    const size_t got = fread(h, 1, 18, f);
    if (got == 0) return 0;
    if (got < 18 || h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return -1;
    const uint16_t xlen = rd16(h + 10);
    std::vector<uint8_t> member((size_t)12 + xlen);
    std::memcpy(member.data(), h, std::min<size_t>(18, member.size()));
    if (member.size() > 18 && fread(member.data() + 18, 1, member.size() - 18, f) != member.size() - 18) return -1;
    int bsize = -1;
    for (size_t x = 0; x + 4 <= xlen;) {
        const uint8_t *sf = member.data() + 12 + x;
        const uint16_t slen = rd16(sf + 2);
        if (sf[0] == 'B' && sf[1] == 'C' && slen == 2 && x + 6 <= xlen) bsize = rd16(sf + 4);
        x += 4 + (size_t)slen;
    }
    if (bsize < 0) return -1;
    const size_t clen = (size_t)bsize + 1, hdr = (size_t)12 + xlen;
    if (clen < hdr + 8) return -1;
    std::vector<uint8_t> rest(clen - hdr);
    // (the header bytes beyond 12 + xlen that were read with the first 18 belong to the payload)
    if (fseeko(f, (off_t)(coff + hdr), SEEK_SET) != 0 || fread(rest.data(), 1, rest.size(), f) != rest.size()) return -1;
    const uint32_t isize = rd32(rest.data() + rest.size() - 4), crc = rd32(rest.data() + rest.size() - 8);
    if (isize > (1u << 16)) return -1;
    const size_t at = out.size();
    out.resize(at + isize);
    if (isize && raw_inflate(rest.data(), rest.size() - 8, out.data() + at, isize, crc) != 0) return -1;
    return (long)clen;
}

void reg2bins(int64_t beg, int64_t end, std::vector<uint32_t> &bins) {
    --end;
    bins.push_back(0);
    for (int64_t k = 1 + (beg >> 26); k <= 1 + (end >> 26); ++k) bins.push_back((uint32_t)k);
    for (int64_t k = 9 + (beg >> 23); k <= 9 + (end >> 23); ++k) bins.push_back((uint32_t)k);
    for (int64_t k = 73 + (beg >> 20); k <= 73 + (end >> 20); ++k) bins.push_back((uint32_t)k);
    for (int64_t k = 585 + (beg >> 17); k <= 585 + (end >> 17); ++k) bins.push_back((uint32_t)k);
    for (int64_t k = 4681 + (beg >> 14); k <= 4681 + (end >> 14); ++k) bins.push_back((uint32_t)k);
}

// the BAM header from the leading BGZF members of `f`: as many as it takes
int read_header_members(Bam &bam, FILE *f, uint32_t &n_ref) {
    std::vector<uint8_t> hbuf;
    uint64_t coff = 0;
    for (;;) {
        const long cl = read_member(f, coff, hbuf);
        if (cl <= 0) return fail(cl == 0 ? "truncated BAM header" : "not a BGZF file (bad gzip member header)");
        coff += (uint64_t)cl;
        const uint8_t *p = hbuf.data();
        bool more = false;
        if (parse_header(bam, p, hbuf.data() + hbuf.size(), n_ref, &more) == 0) return 0;
        if (!more) return -1;
    }
}

// A requested region by reference id (sorted, overlapping / adjacent ones merged).
struct RegionSpan { int32_t tid; int64_t s, e; };

// Regions (reference name, 0-based half-open) -> merged regions by reference id and the chunks of virtual offsets
// that hold every record overlapping one of them: the bins of each region (reg2bins), clipped by the 16 kb linear
// index, sorted and merged (SAM specification section 5; what hts_itr_query does per region, hts.c:1924-1960).
int resolve_regions(const Bam &bam, uint32_t n_ref, int nreg, const char *const *rname, const int64_t *rstart, const int64_t *rend,
                    std::vector<BaiRef> &refs, std::vector<RegionSpan> &merged, std::vector<std::pair<uint64_t, uint64_t>> &chunks) {
    if (load_bai(bam.path, refs) != 0) return -1;
    if (refs.size() != (size_t)n_ref) return fail("the index does not belong to this BAM file (reference count differs): " + bam.path);
    std::vector<RegionSpan> regs;
    for (int i = 0; i < nreg; ++i) {
        int32_t tid = -1;
        for (uint32_t r = 0; r < n_ref; ++r)
            if (bam.ref_names[r] == rname[i]) { tid = (int32_t)r; break; }
        if (tid < 0) continue;                           // unknown chromosome: fetch() finds nothing there
        const int64_t s = std::max<int64_t>(rstart[i], 0), e = std::min<int64_t>(rend[i], (int64_t)1 << 29);
        if (e > s) regs.push_back({tid, s, e});
    }
    std::sort(regs.begin(), regs.end(), [](const RegionSpan &a, const RegionSpan &b) { return a.tid != b.tid ? a.tid < b.tid : (a.s != b.s ? a.s < b.s : a.e < b.e); });
    merged.clear();
    for (const RegionSpan &r : regs) {
        if (!merged.empty() && merged.back().tid == r.tid && r.s <= merged.back().e) merged.back().e = std::max(merged.back().e, r.e);
        else merged.push_back(r);
    }
    std::vector<std::pair<uint64_t, uint64_t>> ch;
    std::vector<uint32_t> bins;
    for (const RegionSpan &r : merged) {
        const BaiRef &br = refs[(size_t)r.tid];
        const size_t w = (size_t)(r.s >> 14);
        if (br.linear.empty()) continue;
        const uint64_t min_off = w < br.linear.size() ? br.linear[w] : br.linear.back();
        bins.clear();
        reg2bins(r.s, r.e, bins);
        std::sort(bins.begin(), bins.end());
        for (size_t b = 0; b < br.bin_id.size(); ++b) {
            if (!std::binary_search(bins.begin(), bins.end(), br.bin_id[b])) continue;
            for (const auto &c : br.chunks[b])
                if (c.second > min_off && c.second > c.first) ch.push_back(c);
        }
    }
    std::sort(ch.begin(), ch.end());
    chunks.clear();
    for (const auto &c : ch) {
        if (!chunks.empty() && c.first <= chunks.back().second) chunks.back().second = std::max(chunks.back().second, c.second);
        else chunks.push_back(c);
    }
    return 0;
}

int decode_regions(Bam &bam, int nthreads, int nreg, const char *const *rname, const int64_t *rstart, const int64_t *rend) {
    Lap lap;
    FILE *f = fopen(bam.path.c_str(), "rb");
    if (!f) return fail("cannot read " + bam.path);
    struct Closer { FILE *f; ~Closer() { if (f) fclose(f); } } closer{f};
    // ---- header: inflate members from the start of the file until it is complete
    uint32_t n_ref = 0;
    if (read_header_members(bam, f, n_ref) != 0) return -1;
    std::vector<BaiRef> refs;
    std::vector<RegionSpan> merged;
    std::vector<std::pair<uint64_t, uint64_t>> chunks;
    if (resolve_regions(bam, n_ref, nreg, rname, rstart, rend, refs, merged, chunks) != 0) return -1;
    lap("header + index");
    // ---- read and inflate what the chunks touch (threads over chunks, one file handle each)
    std::vector<std::vector<uint8_t>> bytes(chunks.size());
    std::atomic<size_t> next(0);
    std::atomic<int> bad(0);
    auto worker = [&]() {
        FILE *g = fopen(bam.path.c_str(), "rb");
        if (!g) { bad.store(1); return; }
        std::vector<uint8_t> buf;
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= chunks.size()) break;
            const uint64_t vb = chunks[k].first, ve = chunks[k].second;
            const uint64_t cb = vb >> 16, ce = ve >> 16;
            const size_t ub = (size_t)(vb & 0xffff), ue = (size_t)(ve & 0xffff);
            buf.clear();
            uint64_t cur = cb;
            size_t stop = (size_t)-1;
            bool ok = true;
            while (cur < ce || (cur == ce && ue > 0)) {
                const size_t before = buf.size();
                const long cl = read_member(g, cur, buf);
                if (cl <= 0) { ok = false; break; }
                if (cur == ce) { stop = before + ue; break; }
                cur += (uint64_t)cl;
            }
            if (stop == (size_t)-1) stop = buf.size();
            if (!ok || ub > stop || stop > buf.size()) { bad.store(2); continue; }
            bytes[k].assign(buf.begin() + (long)ub, buf.begin() + (long)stop);
        }
        fclose(g);
    };
    {
        std::vector<std::thread> pool;
        const int nt = (int)std::min<size_t>((size_t)std::max(nthreads, 1), std::max<size_t>(chunks.size(), 1));
        for (int t = 1; t < nt; ++t) pool.emplace_back(worker);
        worker();
        for (auto &t : pool) t.join();
    }
    if (bad.load() != 0) return fail("damaged BGZF member or index out of step with " + bam.path);
    size_t total = 0;
    for (const auto &b : bytes) total += b.size();
    std::vector<uint8_t> data;
    data.reserve(total);
    for (const auto &b : bytes) data.insert(data.end(), b.begin(), b.end());
    bytes.clear();
    lap("read + inflate chunks");
    if (decode_records(bam, data.data(), data.data() + data.size(), n_ref, nthreads, lap) != 0) return -1;
    // ---- keep what overlaps a requested region (htslib: pos < end && endpos > start)
    std::vector<size_t> first_reg((size_t)n_ref + 1, merged.size());
    for (size_t i = merged.size(); i-- > 0;) first_reg[(size_t)merged[i].tid] = i;
    auto overlaps = [&](int32_t tid, int64_t pos, int64_t endpos) {
        size_t lo = first_reg[(size_t)tid], hi = lo;
        while (hi < merged.size() && merged[hi].tid == tid) ++hi;
        // first region of this reference whose end lies beyond pos
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            if (merged[mid].e <= pos) lo = mid + 1; else hi = mid;
        }
        return lo < merged.size() && merged[lo].tid == tid && merged[lo].s < endpos;
    };
    for (Part &pt : bam.parts) {
        size_t w = 0, rw = 0, rr = 0, wk = 0, ww = 0;   // wk / ww: wide records seen / kept
        for (size_t i = 0; i < pt.n; ++i) {
            int64_t nb = pt.nblk[i], L = pt.alen[i];
            const bool wide = wk < pt.wide_idx.size() && pt.wide_idx[wk] == (int64_t)i;   // true values aside
            if (wide) { nb = pt.wide_nblk[wk]; L = pt.wide_alen[wk]; }
            const size_t runs = nb >= 2 ? (size_t)nb : 0;
            const int64_t endpos = nb >= 2 ? (int64_t)pt.blk_start[rr + runs - 1] + pt.blk_len[rr + runs - 1]
                                           : (int64_t)pt.pos[i] + std::max<int64_t>(L, 1);
            if (overlaps(pt.tid[i], pt.pos[i], endpos)) {
                pt.tid[w] = pt.tid[i]; pt.pos[w] = pt.pos[i]; pt.alen[w] = pt.alen[i]; pt.flags[w] = pt.flags[i]; pt.nblk[w] = pt.nblk[i];
                pt.flag16[w] = pt.flag16[i]; pt.mapq[w] = pt.mapq[i]; pt.lseq[w] = pt.lseq[i]; pt.nh[w] = pt.nh[i];
                for (size_t k = 0; k < runs; ++k) { pt.blk_start[rw + k] = pt.blk_start[rr + k]; pt.blk_len[rw + k] = pt.blk_len[rr + k]; }
                if (wide) { pt.wide_idx[ww] = (int64_t)w; pt.wide_alen[ww] = pt.wide_alen[wk]; pt.wide_nblk[ww] = pt.wide_nblk[wk]; ++ww; }
                ++w;
                rw += runs;
            }
            if (wide) ++wk;
            rr += runs;
        }
        pt.n = w;
        pt.nrun = rw;
        pt.wide_idx.resize(ww); pt.wide_alen.resize(ww); pt.wide_nblk.resize(ww);
    }
    for (size_t k = 0; k < bam.parts.size(); ++k) {
        bam.rec_off[k + 1] = bam.rec_off[k] + bam.parts[k].n;
        bam.run_off[k + 1] = bam.run_off[k] + bam.parts[k].nrun;
    }
    bam.nrec = bam.parts.empty() ? 0 : bam.rec_off[bam.parts.size()];
    bam.nrun = bam.parts.empty() ? 0 : bam.run_off[bam.parts.size()];
    // `mapped` as pysam reports it: from the index, for the whole file
    bool any_meta = false;
    int64_t mapped = 0;
    for (const BaiRef &br : refs) { any_meta |= br.has_meta; mapped += (int64_t)br.n_mapped; }
    bam.mapped = any_meta ? mapped : -1;
    lap("filter");
    bam.loaded = true;
    return 0;
}

// CPUs this process may actually use: the smaller of the hardware threads, the affinity mask and the
// container's CFS quota (cgroup v2 cpu.max / v1 cpu.cfs_quota_us).  Pools sized beyond the quota only
// burn it in bursts and are then throttled as a whole.
static int usable_cpus() {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min<unsigned>(n, (unsigned)std::max(1, CPU_COUNT(&set)));
    long long quota = -1, period = -1;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        if (fscanf(f, "%31s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else {
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = -1; fclose(g); }
    }
    if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
    return (int)n;
}

// one thread per usable CPU, at most 128 (starting hundreds of threads costs more than they return)
int default_threads() { return std::min(128, usable_cpus()); }

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
    if (nthreads <= 0) nthreads = default_threads();
    return decode(*b, nthreads);
}

// region-limited load through the BAI index: only records that overlap one of the `nreg` regions
// (reference name, 0-based half-open) are staged; counts[2] (`mapped`) then comes from the index (-1 if
// the index carries no per-reference counts)
int pb_load_regions(void *h, int nthreads, int nreg, const char *const *names, const int64_t *start, const int64_t *end) {
    Bam *b = static_cast<Bam *>(h);
    if (!b || nreg < 0 || (nreg > 0 && (!names || !start || !end))) return fail("pb_load_regions: bad arguments");
    if (b->loaded) return fail("pb_load_regions: file already loaded");
    if (nthreads <= 0) nthreads = default_threads();
    return decode_regions(*b, nthreads, nreg, names, start, end);
}

// What a decoder that reads the file itself (pc_bam_open_span: BGZF inflate and record decode on the GPU) needs to know
// about a set of regions: the merged regions by reference id (m_tid / m_start / m_end, capacity nreg; *nmerged of them)
// and the SPAN of virtual offsets [*voff_begin, *voff_end) that holds every chunk of every region (0, 0: no record
// overlaps any of them); *mapped = the index's mapped-read count of the whole file (-1: the index carries none).  The
// header is read as a side effect (pb_nref / pb_ref_name / pb_ref_length answer afterwards).
int pb_resolve_regions(void *h, int nreg, const char *const *names, const int64_t *start, const int64_t *end, uint64_t *voff_begin,
                       uint64_t *voff_end, int64_t *mapped, int *nmerged, int32_t *m_tid, int64_t *m_start, int64_t *m_end) {
    Bam *b = static_cast<Bam *>(h);
    if (!b || nreg < 0 || (nreg > 0 && (!names || !start || !end)) || !voff_begin || !voff_end || !mapped || !nmerged)
        return fail("pb_resolve_regions: bad arguments");
    FILE *f = fopen(b->path.c_str(), "rb");
    if (!f) return fail("cannot read " + b->path);
    uint32_t n_ref = 0;
    const int hrc = read_header_members(*b, f, n_ref);
    fclose(f);
    if (hrc != 0) return -1;
    std::vector<BaiRef> refs;
    std::vector<RegionSpan> merged;
    std::vector<std::pair<uint64_t, uint64_t>> chunks;
    if (resolve_regions(*b, n_ref, nreg, names, start, end, refs, merged, chunks) != 0) return -1;
    *voff_begin = chunks.empty() ? 0 : chunks.front().first;
    *voff_end = 0;
    for (const auto &c : chunks) *voff_end = std::max(*voff_end, c.second);
    int64_t m = 0;
    bool any_meta = false;
    for (const BaiRef &br : refs) { m += (int64_t)br.n_mapped; any_meta |= br.has_meta; }
    *mapped = any_meta ? m : -1;
    *nmerged = (int)merged.size();
    for (size_t k = 0; k < merged.size() && m_tid && m_start && m_end; ++k) { m_tid[k] = merged[k].tid; m_start[k] = merged[k].s; m_end[k] = merged[k].e; }
    return 0;
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

// wide records (see Part): how many, and their flat record indices / true aligned lengths / run counts
int64_t pb_wide_count(void *h) {
    Bam *b = static_cast<Bam *>(h);
    if (!b || !b->loaded) return -1;
    int64_t n = 0;
    for (const Part &pt : b->parts) n += (int64_t)pt.wide_idx.size();
    return n;
}

int pb_fill_wide(void *h, int64_t *idx, int32_t *alen, int32_t *nblk) {
    Bam *b = static_cast<Bam *>(h);
    if (!b || !b->loaded) return fail("pb_fill_wide: file not loaded");
    size_t at = 0;
    for (size_t k = 0; k < b->parts.size(); ++k) {
        const Part &pt = b->parts[k];
        for (size_t j = 0; j < pt.wide_idx.size(); ++j, ++at) {
            idx[at] = (int64_t)b->rec_off[k] + pt.wide_idx[j];
            alen[at] = pt.wide_alen[j];
            nblk[at] = pt.wide_nblk[j];
        }
    }
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
            const size_t at = b->rec_off[k], n = pt.n, rat = b->run_off[k], m = pt.nrun;
            if (n) {
                std::memcpy(tid + at, pt.tid, n * 4);
                std::memcpy(pos + at, pt.pos, n * 4);
                std::memcpy(alen + at, pt.alen, n * 2);
                std::memcpy(flags + at, pt.flags, n);
                std::memcpy(nblk + at, pt.nblk, n);
            }
            if (m) {
                std::memcpy(blk_start + rat, pt.blk_start, m * 4);
                std::memcpy(blk_len + rat, pt.blk_len, m * 4);
            }
        }
    };
    std::vector<std::thread> pool;
    const int nt = (int)std::min<size_t>((size_t)std::min(std::max(b->threads, 1), 32), std::max<size_t>(nparts, 1));   // a copy: memory-bound
    for (int t = 1; t < nt; ++t) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    return 0;
}


// The SAM FLAG word, MAPQ and l_seq of every staged record (same order as pb_fill's columns): what pysam exposes as
// read.flag / .mapping_quality / .query_length (and the is_* properties derived from the flag bits) to the filter
// functions of BAMGenomeArray.add_filter (genome_array.py:697-722).  Any pointer may be NULL.
// The NH:i tag of every staged record (same order): its value clamped to 65 535, 0 where a record has none -- what
// `read.get_tag("NH")` / `read.has_tag("NH")` answer from in a filter function (genome_array.py:697-722, 819-820).
int pb_fill_nh(void *h, uint16_t *nh) {
    Bam *b = static_cast<Bam *>(h);
    if (!b || !b->loaded) return fail("pb_fill_nh: file not loaded");
    for (size_t k = 0; k < b->parts.size(); ++k) {
        const Part &pt = b->parts[k];
        if (pt.n && nh) std::memcpy(nh + b->rec_off[k], pt.nh, pt.n * 2);
    }
    return 0;
}

int pb_fill_sam(void *h, uint16_t *flag16, uint8_t *mapq, int32_t *lseq) {
    Bam *b = static_cast<Bam *>(h);
    if (!b || !b->loaded) return fail("pb_fill_sam: file not loaded");
    for (size_t k = 0; k < b->parts.size(); ++k) {
        const Part &pt = b->parts[k];
        const size_t at = b->rec_off[k], n = pt.n;
        if (!n) continue;
        if (flag16) std::memcpy(flag16 + at, pt.flag16, n * 2);
        if (mapq) std::memcpy(mapq + at, pt.mapq, n);
        if (lseq) std::memcpy(lseq + at, pt.lseq, n * 4);
    }
    return 0;
}

} // extern "C"
