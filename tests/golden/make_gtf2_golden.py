"""Golden vectors for the GTF2 -> transcript assembly step.

Run in THIS container against the scratch build of the reference
(tests/golden/build_scratch_reference.sh):

    PYTHONPATH=/tmp/oracle:/tmp/oracle/stubs python tests/golden/make_gtf2_golden.py

Writes tests/golden/gtf2_transcripts.json = {"gtf2": <input text>, "transcripts":
[[name, "chrom:s-e^s-e(strand)", cds_genome_start, cds_genome_end], ...]} in the order
``GTF2_TranscriptAssembler`` (plastid/readers/gff.py:1007-1206) yields them.
The input text is synthetic (seeded), written by this script.
"""
import io
import json
import os
import random
import warnings

from plastid.readers.gff import GTF2_TranscriptAssembler


def synth_gtf2(seed=7, ngenes=60):
    rnd = random.Random(seed)
    lines = ["# synthetic GTF2", ""]
    for g in range(ngenes):
        chrom = rnd.choice(["chrI", "chrII", "chrM", "2-micron"])
        strand = rnd.choice("+-")
        gstart = rnd.randrange(1, 200000)
        for t in range(rnd.randrange(1, 4)):
            tname = "g%03d.t%d" % (g, t)
            nex = rnd.randrange(1, 7)
            pos, exons = gstart + rnd.randrange(0, 50), []
            for _ in range(nex):
                ln = rnd.randrange(20, 400)
                exons.append((pos, pos + ln - 1))
                pos += ln + rnd.choice([0, 1, 1, 30, 500])  # 0 -> overlapping by 1?, 1 -> adjacent
            attrs = 'gene_id "g%03d"; transcript_id "%s"; tag "a;b";' % (g, tname)
            mode = rnd.choice(["exon", "exon+cds", "cds_only", "utr"])
            feats = []
            if mode in ("exon", "exon+cds", "utr"):
                feats += [("exon", s, e) for s, e in exons]
            if mode in ("exon+cds", "cds_only"):
                cs = exons[0][0] + rnd.randrange(0, 10)
                ce = exons[-1][1] - rnd.randrange(0, 10)
                for s, e in exons:
                    s2, e2 = max(s, cs), min(e, ce)
                    if s2 <= e2:
                        feats.append(("CDS", s2, e2))
                if mode == "cds_only":
                    feats.append(("stop_codon", ce + 1, ce + 3))
                    feats.append(("start_codon", cs, cs + 2))
            if mode == "utr":
                feats.append(("5UTR", exons[0][0] - 10, exons[0][0] + 5))
                feats.append(("3UTR", exons[-1][1] + 1, exons[-1][1] + 40))
            feats.append(("gene", exons[0][0], exons[-1][1]))        # ignored type
            feats.append(("transcript", exons[0][0], exons[-1][1]))  # ignored type
            rnd.shuffle(feats)
            for ftype, s, e in feats:
                extra = ' exon_number "%d";' % rnd.randrange(1, 9) if ftype == "exon" else ""
                lines.append("\t".join([chrom, "synth", ftype, str(s), str(e), ".", strand, ".", attrs + extra]))
    # a transcript on two strands and one on two chromosomes: both rejected
    lines.append("\t".join(["chrI", "synth", "exon", "100", "200", ".", "+", ".", 'gene_id "bad"; transcript_id "bad.1";']))
    lines.append("\t".join(["chrI", "synth", "exon", "300", "400", ".", "-", ".", 'gene_id "bad"; transcript_id "bad.1";']))
    lines.append("\t".join(["chrI", "synth", "exon", "100", "200", ".", "+", ".", 'gene_id "bad"; transcript_id "bad.2";']))
    lines.append("\t".join(["chrII", "synth", "exon", "300", "400", ".", "+", ".", 'gene_id "bad"; transcript_id "bad.2";']))
    # identical span/length: ordering falls through to the name
    for nm in ("tie.b", "tie.a"):
        lines.append("\t".join(["chrI", "synth", "exon", "5000", "5100", ".", "+", ".",
                                'gene_id "tie"; transcript_id "%s"' % nm]))
    # percent escapes (unescape_GTF2, gff_tokens.py:582-599) in ids, a repeated key (joined with a comma, with
    # a FileFormatWarning), a lower-case escape and one outside the reference's table (both stay literal)
    for s, e in ((7000, 7100), (7200, 7300)):
        lines.append("\t".join(["chrI", "synth", "exon", str(s), str(e), ".", "+", ".",
                                'gene_id "esc%2Cg"; transcript_id "esc%3B1"; note "q%22uoted%25"; note "second";']))
    lines.append("\t".join(["chrI", "synth", "exon", "8000", "8100", ".", "+", ".", 'gene_id "low"; transcript_id "low%3b%41"']))
    return "\n".join(lines) + "\n"


def main():
    text = synth_gtf2()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        txs = list(GTF2_TranscriptAssembler(io.StringIO(text)))
    rejected = sorted(str(x.message).split("'")[1] for x in w if "Rejecting" in str(x.message))
    rows = [[t.get_name(), str(t), t.attr.get("cds_genome_start"), t.attr.get("cds_genome_end"),
             t.attr.get("gene_id")] for t in txs]
    notes = {t.get_name(): t.attr.get("note") for t in txs if t.attr.get("note") is not None}
    n_dup = sum(1 for x in w if "duplicate attribute key" in str(x.message))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gtf2_transcripts.json")
    with open(out, "w") as fh:
        json.dump({"gtf2": text, "transcripts": rows, "rejected": rejected, "notes": notes, "duplicate_key_warnings": n_dup}, fh)
    print(len(rows), "transcripts,", len(rejected), "rejected ->", out)


if __name__ == "__main__":
    main()
