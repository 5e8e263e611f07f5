/*
 * hts_golden.c -- fixture generator that runs the REFERENCE-HELD htslib (the copy vendored under
 * /root/reference/kent/src/htslib, version 1.3) in the build container.  Test infrastructure:
 * built by oracle/build_ref.sh into oracle/_ref/ (never committed), run by
 * tests/golden/make_hts_golden.py, whose outputs (BAM + BAI bytes and the expected arrays) are the
 * committed fixture tests/golden/hts_fixture.npz.
 *
 * What it pins (everything the native reader, packing.cigar_to_runs and the oracle's
 * po_cigar_to_runs restate from the SAM specification):
 *   - the BAM / BGZF / BAI bytes htslib itself writes for seeded records over all nine CIGAR
 *     operations (sam_write1, sam_index_build);
 *   - per record, as htslib reads it back: pos, flag, MAPQ, l_qseq, bam_endpos() (sam.c:329-341) and the aligned
 *     reference positions from a CIGAR walk driven by htslib's own bam_cigar_type() table
 *     (htslib/sam.h:64-104: an op that consumes query AND reference emits positions) -- pysam's
 *     AlignedSegment.positions;
 *   - the result set of hts_itr_query / sam_itr_next for seeded regions (hts.c:1924-1960, the
 *     fetch the reference calls at genome_array.py:800-809);
 *   - the per-reference mapped / unmapped counts of the index (pysam AlignmentFile.mapped).
 *
 * Output (stdout), one line per item:
 *   REF name length
 *   REC index tid pos flag endpos npos p0 p1 ...
 *   SAM index mapq l_qseq                      (core.qual / core.l_qseq as read back: pysam's mapping_quality / query_length)
 *   AUX index has nh                           (bam_aux_get(b, "NH") != NULL and bam_aux2i of it: pysam's has_tag / get_tag;
 *                                               a third of the records carry the tag -- every integer type, behind other
 *                                               fields of every kind -- and some an NH of a non-integer type)
 *   CIG index n op0 len0 op1 len1 ...          (BAM op codes 0..8 = MIDNSHP=X, as read back)
 *   REG tid beg end n i0 i1 ...
 *   STAT tid mapped unmapped
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "htslib/hts.h"
#include "htslib/sam.h"

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint32_t rnd(void) { /* xorshift64* */
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return (uint32_t)((rng_state * 0x2545F4914F6CDD1Dull) >> 32);
}
static int rint_(int lo, int hi) { return lo + (int)(rnd() % (uint32_t)(hi - lo + 1)); }

#define MAXOPS 64

/* a valid CIGAR: [H][S] M-like (op M-like)* [S][H]; between two M-like blocks one of I D N P or
 * nothing (adjacent M/=/X blocks) */
static int make_cigar(uint32_t *cig, int style) {
    int n = 0;
    const int mlike[3] = {BAM_CMATCH, BAM_CEQUAL, BAM_CDIFF};
    if (style == 0) { cig[n++] = bam_cigar_gen(rint_(20, 40), BAM_CMATCH); return n; }   /* plain <L>M */
    if (rnd() % 4 == 0) cig[n++] = bam_cigar_gen(rint_(1, 5), BAM_CHARD_CLIP);
    if (rnd() % 3 == 0) cig[n++] = bam_cigar_gen(rint_(1, 8), BAM_CSOFT_CLIP);
    const int nblocks = style == 1 ? rint_(1, 3) : rint_(2, 6);
    for (int b = 0; b < nblocks; ++b) {
        cig[n++] = bam_cigar_gen(rint_(1, 25), mlike[rnd() % 3]);
        if (b + 1 < nblocks) {
            switch (rnd() % 6) {
            case 0: cig[n++] = bam_cigar_gen(rint_(1, 4), BAM_CINS); break;
            case 1: cig[n++] = bam_cigar_gen(rint_(1, 6), BAM_CDEL); break;
            case 2: cig[n++] = bam_cigar_gen(style == 2 ? rint_(50, 30000) : rint_(20, 400), BAM_CREF_SKIP); break;
            case 3: cig[n++] = bam_cigar_gen(rint_(1, 3), BAM_CPAD); break;
            case 4: cig[n++] = bam_cigar_gen(rint_(1, 3), BAM_CDEL); cig[n++] = bam_cigar_gen(rint_(1, 3), BAM_CINS); break;
            default: break; /* adjacent aligned blocks (e.g. 5M3X4=) */
            }
        }
    }
    if (rnd() % 3 == 0) cig[n++] = bam_cigar_gen(rint_(1, 8), BAM_CSOFT_CLIP);
    if (rnd() % 4 == 0) cig[n++] = bam_cigar_gen(rint_(1, 5), BAM_CHARD_CLIP);
    return n;
}

static void fill(bam1_t *b, int idx, int tid, int pos, int flag, const uint32_t *cig, int ncig) {
    char name[32];
    snprintf(name, sizeof(name), "r%d", idx);
    int lq = 0;
    for (int k = 0; k < ncig; ++k)
        if (bam_cigar_type(bam_cigar_op(cig[k])) & 1) lq += bam_cigar_oplen(cig[k]);
    if (ncig == 0) lq = 12;
    const int l_qname = (int)strlen(name) + 1;
    const int need = l_qname + 4 * ncig + (lq + 1) / 2 + lq;
    if ((int)b->m_data < need) { b->m_data = need; b->data = (uint8_t *)realloc(b->data, (size_t)need); }
    b->l_data = need;
    memset(b->data, 0, (size_t)need);
    memcpy(b->data, name, (size_t)l_qname);
    memcpy(b->data + l_qname, cig, 4u * (size_t)ncig);
    memset(b->data + l_qname + 4 * ncig, 0x11, (size_t)((lq + 1) / 2));
    memset(b->data + l_qname + 4 * ncig + (lq + 1) / 2, 30, (size_t)lq);
    /* MAPQ varies with the record index (0 .. 60, every 17th record 255 = "not available"); no random draw, so every
     * other field of the fixture is what it was before the column was added */
    b->core.tid = tid; b->core.pos = pos; b->core.qual = (uint8_t)(idx % 17 == 16 ? 255 : (idx * 37 + 11) % 61); b->core.l_qname = (uint8_t)l_qname;
    b->core.flag = (uint16_t)flag; b->core.n_cigar = (uint16_t)ncig; b->core.l_qseq = lq;
    b->core.mtid = -1; b->core.mpos = -1; b->core.isize = 0;
    b->core.bin = hts_reg2bin(pos, bam_endpos(b), 14, 5);
}

/* auxiliary fields by record index alone (no random draw: every other field of the fixture stays what it was): an NH tag
 * on every third record, in turn of every integer type and value range, in front of / behind / between other fields of
 * every type (A c s i f Z H B); every 31st record an NH of type Z or f (not an integer: pysam's get_tag would hand back
 * the string / float; bam_aux2i answers 0) */
static void add_aux(bam1_t *b, int idx) {
    if (idx % 3 != 0 && idx % 31 != 5) {
        if (idx % 7 == 1) { int32_t nm = idx % 5; bam_aux_append(b, "NM", 'i', 4, (uint8_t *)&nm); }
        return;
    }
    const int k = idx / 3;
    if (k % 2 == 0) { char xs = (k & 2) ? '+' : '-'; bam_aux_append(b, "XS", 'A', 1, (uint8_t *)&xs); }
    if (k % 3 == 0) { char md[16]; snprintf(md, sizeof(md), "%d", 10 + k % 30); bam_aux_append(b, "MD", 'Z', (int)strlen(md) + 1, (uint8_t *)md); }
    if (k % 5 == 0) { uint8_t arr[5 + 3 * 2] = {'S', 3, 0, 0, 0, 1, 0, 2, 0, 3, 0}; bam_aux_append(b, "ZB", 'B', (int)sizeof(arr), arr); }
    if (k % 4 == 1) { float f = 0.5f * (float)(k % 9); bam_aux_append(b, "XF", 'f', 4, (uint8_t *)&f); }
    if (k % 6 == 2) { char hx[8] = "1AE3"; bam_aux_append(b, "XH", 'H', (int)strlen(hx) + 1, (uint8_t *)hx); }
    if (idx % 31 == 5 && idx % 3 != 0) {      /* an NH that is no integer */
        if (idx % 2) { char z[4] = "2"; bam_aux_append(b, "NH", 'Z', 2, (uint8_t *)z); }
        else { float f = 3.0f; bam_aux_append(b, "NH", 'f', 4, (uint8_t *)&f); }
    } else {
        switch (k % 6) {
        case 0: { uint8_t v = (uint8_t)(1 + k % 4); bam_aux_append(b, "NH", 'C', 1, &v); break; }
        case 1: { int8_t v = (int8_t)((k % 8 == 1) ? -3 : 2); bam_aux_append(b, "NH", 'c', 1, (uint8_t *)&v); break; }
        case 2: { uint16_t v = (uint16_t)(k % 10 == 2 ? 65535 : 300 + k % 50); bam_aux_append(b, "NH", 'S', 2, (uint8_t *)&v); break; }
        case 3: { int16_t v = (int16_t)(k % 12 == 3 ? -1 : 1); bam_aux_append(b, "NH", 's', 2, (uint8_t *)&v); break; }
        case 4: { int32_t v = (k % 10 == 4) ? 70000 : 1; bam_aux_append(b, "NH", 'i', 4, (uint8_t *)&v); break; }
        default: { uint32_t v = (k % 18 == 5) ? 4000000000u : 5; bam_aux_append(b, "NH", 'I', 4, (uint8_t *)&v); break; }
        }
    }
    if (k % 2 == 1) { int16_t as = (int16_t)(-k % 90); bam_aux_append(b, "AS", 's', 2, (uint8_t *)&as); }
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: hts_golden out.bam nrecords [seed]\n"); return 2; }
    const char *fn = argv[1];
    const int nrec = atoi(argv[2]);
    if (argc > 3) rng_state ^= (uint64_t)atoll(argv[3]) * 0x9E3779B97F4A7C15ull;
    const char *names[3] = {"chrA", "chrB", "chrC"};
    const int lens[3] = {400000, 60000, 2000};

    bam_hdr_t *h = bam_hdr_init();
    h->n_targets = 3;
    h->target_len = (uint32_t *)malloc(3 * sizeof(uint32_t));
    h->target_name = (char **)malloc(3 * sizeof(char *));
    char text[256] = "@HD\tVN:1.5\tSO:coordinate\n";
    for (int t = 0; t < 3; ++t) {
        h->target_len[t] = (uint32_t)lens[t];
        h->target_name[t] = strdup(names[t]);
        char line[64];
        snprintf(line, sizeof(line), "@SQ\tSN:%s\tLN:%d\n", names[t], lens[t]);
        strcat(text, line);
        printf("REF %s %d\n", names[t], lens[t]);
    }
    h->text = strdup(text);
    h->l_text = (uint32_t)strlen(text);

    samFile *out = sam_open(fn, "wb");
    if (!out || sam_hdr_write(out, h) < 0) { fprintf(stderr, "cannot write %s\n", fn); return 1; }
    bam1_t *b = bam_init1();
    /* records sorted by (tid, pos): per contig a share of the records, positions by random increments
     * (many ties and pile-ups, some long gaps) */
    const int share[3] = {nrec * 7 / 10, nrec * 28 / 100, nrec - nrec * 7 / 10 - nrec * 28 / 100};
    int idx = 0;
    for (int t = 0; t < 3; ++t) {
        int pos = rint_(0, 50);
        for (int k = 0; k < share[t]; ++k, ++idx) {
            const int step = (rnd() % 5 == 0) ? 0 : ((rnd() % 50 == 0) ? rint_(200, 3000) : rint_(0, lens[t] / (share[t] + 1) * 2 + 1));
            pos += step;
            if (pos > lens[t] - 12) pos = lens[t] - 12;               /* monotone: the file stays coordinate sorted */
            uint32_t cig[MAXOPS];
            int style = (int)(rnd() % 10);
            style = style < 5 ? 0 : (style < 8 ? 1 : 2);
            int ncig = make_cigar(cig, style);
            int flag = (rnd() & 1) ? BAM_FREVERSE : 0;
            const uint32_t f = rnd() % 40;
            if (f == 0) flag |= BAM_FSECONDARY;
            if (f == 1) flag |= BAM_FDUP;
            if (f == 2) flag |= BAM_FQCFAIL;
            if (f == 3) { flag |= BAM_FUNMAP; ncig = 0; }        /* placed but unmapped: no CIGAR, endpos = pos + 1 */
            if (f == 4) flag |= BAM_FPAIRED | BAM_FREAD2 | BAM_FMREVERSE;
            fill(b, idx, t, pos, flag, cig, ncig);
            /* keep every alignment inside its contig */
            if (bam_endpos(b) > lens[t]) { const uint32_t one = bam_cigar_gen(rint_(1, 10), BAM_CMATCH); fill(b, idx, t, pos, flag & ~BAM_FUNMAP, &one, 1); }
            add_aux(b, idx);
            if (sam_write1(out, h, b) < 0) { fprintf(stderr, "write failed\n"); return 1; }
        }
    }
    /* two unplaced reads at the end (tid -1): fetch never returns them */
    for (int k = 0; k < 2; ++k, ++idx) {
        fill(b, idx, -1, -1, BAM_FUNMAP, NULL, 0);
        b->core.bin = hts_reg2bin(-1, 0, 14, 5);
        if (sam_write1(out, h, b) < 0) return 1;
    }
    sam_close(out);
    if (sam_index_build(fn, 0) < 0) { fprintf(stderr, "index build failed\n"); return 1; }

    /* ---- read back through htslib */
    samFile *in = sam_open(fn, "rb");
    bam_hdr_t *h2 = sam_hdr_read(in);
    int i = 0;
    while (sam_read1(in, h2, b) >= 0) {
        const uint32_t *cig = bam_get_cigar(b);
        int npos = 0;
        for (int k = 0; k < b->core.n_cigar; ++k)
            if ((bam_cigar_type(bam_cigar_op(cig[k])) & 3) == 3) npos += bam_cigar_oplen(cig[k]);
        printf("REC %d %d %d %d %d %d", i, b->core.tid, b->core.pos, b->core.flag, (int)bam_endpos(b), npos);
        int rp = b->core.pos;
        for (int k = 0; k < b->core.n_cigar; ++k) {
            const int op = bam_cigar_op(cig[k]), len = bam_cigar_oplen(cig[k]), type = bam_cigar_type(op);
            if ((type & 3) == 3) for (int x = 0; x < len; ++x) printf(" %d", rp + x);
            if (type & 2) rp += len;
        }
        printf("\n");
        printf("SAM %d %d %d\n", i, (int)b->core.qual, (int)b->core.l_qseq);     /* pysam: mapping_quality, query_length */
        {
            const uint8_t *a = bam_aux_get(b, "NH");
            printf("AUX %d %d %lld\n", i, a ? 1 : 0, a ? (long long)bam_aux2i(a) : 0ll);
        }
        printf("CIG %d %d", i, b->core.n_cigar);
        for (int k = 0; k < b->core.n_cigar; ++k) printf(" %d %d", bam_cigar_op(cig[k]), bam_cigar_oplen(cig[k]));
        printf("\n");
        ++i;
    }
    sam_close(in);

    in = sam_open(fn, "rb");
    bam_hdr_destroy(sam_hdr_read(in));
    hts_idx_t *ix = sam_index_load(in, fn);
    if (!ix) { fprintf(stderr, "cannot load index\n"); return 1; }
    for (int q = 0; q < 400; ++q) {
        const int t = (int)(rnd() % 3);
        int beg, end;
        switch (q % 5) {
        case 0: beg = rint_(0, lens[t] - 1); end = beg + rint_(1, 50); break;            /* short */
        case 1: beg = rint_(0, lens[t] - 1); end = beg + rint_(100, 20000); break;       /* long */
        case 2: beg = (rint_(0, lens[t]) >> 14) << 14; end = beg + (1 << 14); break;      /* one 16 kb bin exactly */
        case 3: beg = 0; end = lens[t]; break;                                            /* whole contig */
        default: beg = rint_(0, lens[t] - 1); end = beg + 1; break;                       /* one position */
        }
        if (end > lens[t] + 100) end = lens[t] + 100;
        hts_itr_t *it = sam_itr_queryi(ix, t, beg, end);
        int n = 0, cap = 1024, *got = (int *)malloc(sizeof(int) * (size_t)cap);
        while (it && sam_itr_next(in, it, b) >= 0) {
            if (n == cap) { cap *= 2; got = (int *)realloc(got, sizeof(int) * (size_t)cap); }
            got[n++] = atoi(bam_get_qname(b) + 1);
        }
        printf("REG %d %d %d %d", t, beg, end, n);
        for (int k = 0; k < n; ++k) printf(" %d", got[k]);
        printf("\n");
        free(got);
        hts_itr_destroy(it);
    }
    for (int t = 0; t < 3; ++t) {
        uint64_t m = 0, u = 0;
        hts_idx_get_stat(ix, t, &m, &u);
        printf("STAT %d %llu %llu\n", t, (unsigned long long)m, (unsigned long long)u);
    }
    printf("NOCOOR %llu\n", (unsigned long long)hts_idx_get_n_no_coor(ix));
    hts_idx_destroy(ix);
    sam_close(in);
    bam_destroy1(b);
    return 0;
}
