"""Golden vectors for GFF3 -> transcript assembly.

Run in THIS container against the scratch build of the reference:

    PYTHONPATH=/tmp/oracle:/tmp/oracle/stubs python tests/golden/make_gff3_golden.py

Writes tests/golden/gff3_transcripts.json = {"gff3": <input text>, "transcripts":
[[name, "chrom:s-e^s-e(strand)", cds_genome_start, cds_genome_end, gene_id, type], ...],
"rejected": [...]} in the order ``GFF3_TranscriptAssembler`` (plastid/readers/gff.py:1216-1575)
yields them.  The input text is synthetic (seeded), written by this script.
"""
import io
import json
import os
import random
import warnings

from plastid.readers.gff import GFF3_TranscriptAssembler


def synth_gff3(seed=11, ngenes=45):
    rnd = random.Random(seed)
    L = ["##gff-version 3", "# synthetic"]

    def row(chrom, ftype, s, e, strand, attrs):
        L.append("\t".join([chrom, "synth", ftype, str(s), str(e), ".", strand, ".", attrs]))

    for g in range(ngenes):
        chrom = rnd.choice(["chrI", "chrII", "chrM"])
        strand = rnd.choice("+-")
        start = rnd.randrange(1, 150000)
        gid = "gene%03d" % g
        row(chrom, "gene", start, start + 5000, strand, "ID=%s;Name=%s%%20x" % (gid, gid))
        style = rnd.choice(["flybase", "flybase", "wormbase", "implied", "noncoding", "childless", "shared"])
        ntx = rnd.randrange(1, 3)
        exons = []
        pos = start
        for _ in range(rnd.randrange(1, 6)):
            ln = rnd.randrange(30, 300)
            exons.append((pos, pos + ln - 1))
            pos += ln + rnd.choice([0, 1, 40, 400])
        for t in range(ntx):
            tid = "%s.t%d" % (gid, t)
            mine = exons if t == 0 else exons[:max(1, len(exons) - 1)]
            if style in ("flybase", "noncoding", "childless", "shared"):
                ttype = "ncRNA" if style == "noncoding" else rnd.choice(["mRNA", "transcript"])
                row(chrom, ttype, mine[0][0], mine[-1][1], strand, "ID=%s;Parent=%s;Note=a%%2Cb,c" % (tid, gid))
            if style == "childless":
                continue
            if style == "flybase" or style == "noncoding":
                for k, (s, e) in enumerate(mine):
                    row(chrom, "exon", s, e, strand, "ID=%s.e%d;Parent=%s" % (tid, k, tid))
                if style == "flybase":
                    cs, ce = mine[0][0] + 5, mine[-1][1] - 5
                    for s, e in mine:
                        s2, e2 = max(s, cs), min(e, ce)
                        if s2 <= e2:
                            row(chrom, "CDS", s2, e2, strand, "ID=%s.cds;Parent=%s" % (tid, tid))
            elif style == "shared" and t == 0:
                # exons that belong to both transcripts of the gene
                parents = ",".join("%s.t%d" % (gid, k) for k in range(ntx))
                for k, (s, e) in enumerate(mine):
                    row(chrom, "exon", s, e, strand, "ID=%s.se%d;Parent=%s" % (gid, k, parents))
            elif style == "wormbase":
                for s, e in mine:      # no transcript feature, no Parent: grouped by shared ID
                    row(chrom, "coding_exon", s, e, strand, "ID=%s;gene=%s" % (tid, gid))
            elif style == "implied":
                for k, (s, e) in enumerate(mine):   # Parent names something that is not a transcript feature
                    row(chrom, "exon", s, e, strand, "ID=%s.x%d;Parent=%s;biotype=protein_coding" % (tid, k, gid))
        if g % 9 == 8:
            L.append("###")
    # exons of one parent on two strands: rejected
    row("chrI", "mRNA", 100, 900, "+", "ID=bad.t;Parent=bad")
    row("chrI", "exon", 100, 200, "+", "ID=bad.e1;Parent=bad.t")
    row("chrI", "exon", 300, 400, "-", "ID=bad.e2;Parent=bad.t")
    # same span and length: order falls through to the name
    for nm in ("tie.b", "tie.a"):
        row("chrII", "mRNA", 7000, 7100, "+", "ID=%s" % nm)
        row("chrII", "exon", 7000, 7100, "+", "Parent=%s" % nm)
    L.append("##FASTA")   # (the reference treats it as one more batch border; it cannot read the sequence lines)
    return "\n".join(L) + "\n"


def main():
    text = synth_gff3()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        txs = list(GFF3_TranscriptAssembler(io.StringIO(text)))
    rejected = sorted(str(x.message).split("'")[1] for x in w if "Rejecting" in str(x.message))
    rows = [[t.get_name(), str(t), t.attr.get("cds_genome_start"), t.attr.get("cds_genome_end"), t.attr.get("gene_id"),
             t.attr.get("type")] for t in txs]
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gff3_transcripts.json")
    with open(out, "w") as fh:
        json.dump({"gff3": text, "transcripts": rows, "rejected": rejected}, fh)
    print(len(rows), "transcripts,", len(rejected), "rejected ->", out)


if __name__ == "__main__":
    main()
