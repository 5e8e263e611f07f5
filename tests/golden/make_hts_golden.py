#!/usr/bin/env python
"""Generate ``tests/golden/hts_fixture.npz`` by running the reference-held htslib (vendored under
``/root/reference/kent/src/htslib``) in the build container.

    python tests/golden/make_hts_golden.py

``oracle/build_ref.sh`` compiles that htslib from the sources where they lie (gcc + system zlib) into
``oracle/_ref/`` together with the harness ``tests/golden/hts_golden.c``; the harness writes a BAM file
and its BAI index for seeded records over all nine CIGAR operations and prints what htslib itself
reads back (see the header of hts_golden.c).  Only DATA is committed: the BAM / BAI bytes and the
expected arrays.  Nothing of the reference travels.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
NREC = 3000
SEED = 20261002


def main():
    subprocess.check_call(["bash", os.path.join(ROOT, "oracle", "build_ref.sh")])
    exe = os.path.join(ROOT, "oracle", "_ref", "hts_golden")
    if not os.path.exists(exe):
        raise SystemExit("oracle/_ref/hts_golden was not built (is /root/reference present?)")
    tmp = tempfile.mkdtemp(prefix="hts_golden_")
    bam = os.path.join(tmp, "g.bam")
    text = subprocess.check_output([exe, bam, str(NREC), str(SEED)]).decode()
    refs, lens = [], []
    rec, pos_off, pos_val = [], [0], []
    cig_off, cig_op, cig_len = [0], [], []
    reg, reg_off, reg_val = [], [0], []
    stat, nocoor = [], 0
    sam = []
    aux = []
    for line in text.splitlines():
        f = line.split()
        if f[0] == "REF":
            refs.append(f[1]); lens.append(int(f[2]))
        elif f[0] == "REC":
            idx, tid, pos, flag, end, npos = (int(x) for x in f[1:7])
            assert idx == len(rec)
            rec.append((tid, pos, flag, end))
            p = [int(x) for x in f[7:]]
            assert len(p) == npos
            pos_val.extend(p); pos_off.append(len(pos_val))
        elif f[0] == "SAM":
            assert int(f[1]) == len(sam)
            sam.append((int(f[2]), int(f[3])))
        elif f[0] == "AUX":
            assert int(f[1]) == len(aux)
            aux.append((int(f[2]), int(f[3])))
        elif f[0] == "CIG":
            n = int(f[2])
            v = [int(x) for x in f[3:]]
            assert len(v) == 2 * n
            cig_op.extend(v[0::2]); cig_len.extend(v[1::2]); cig_off.append(len(cig_op))
        elif f[0] == "REG":
            tid, beg, end, n = (int(x) for x in f[1:5])
            v = [int(x) for x in f[5:]]
            assert len(v) == n
            reg.append((tid, beg, end)); reg_val.extend(v); reg_off.append(len(reg_val))
        elif f[0] == "STAT":
            stat.append((int(f[1]), int(f[2]), int(f[3])))
        elif f[0] == "NOCOOR":
            nocoor = int(f[1])
    rec = np.array(rec, np.int64)
    sam = np.array(sam, np.int64)
    aux = np.array(aux, np.int64)
    assert len(sam) == len(rec) == len(aux)
    out = os.path.join(HERE, "hts_fixture.npz")
    np.savez_compressed(
        out,
        htslib_version=np.array("1.3 (vendored: kent/src/htslib)"),
        bam=np.frombuffer(open(bam, "rb").read(), np.uint8), bai=np.frombuffer(open(bam + ".bai", "rb").read(), np.uint8),
        references=np.array(refs), lengths=np.array(lens, np.int64),
        tid=rec[:, 0].astype(np.int32), pos=rec[:, 1].astype(np.int32), flag=rec[:, 2].astype(np.int32),
        endpos=rec[:, 3].astype(np.int64), mapq=sam[:, 0].astype(np.uint8), l_qseq=sam[:, 1].astype(np.int32),
        has_nh=aux[:, 0].astype(np.uint8), nh=aux[:, 1].astype(np.int64),     # bam_aux_get(b, "NH") != NULL, bam_aux2i of it
        positions_off=np.array(pos_off, np.int64), positions=np.array(pos_val, np.int32),
        cigar_off=np.array(cig_off, np.int64), cigar_op=np.array(cig_op, np.uint8), cigar_len=np.array(cig_len, np.int32),
        regions=np.array(reg, np.int64), region_off=np.array(reg_off, np.int64), region_records=np.array(reg_val, np.int32),
        index_stat=np.array(stat, np.int64), n_no_coor=np.array(nocoor, np.int64))
    print("wrote %s: %d records (%d aligned positions), %d regions, BAM %d B, BAI %d B; %.0f KB" % (
        out, len(rec), len(pos_val), len(reg), os.path.getsize(bam), os.path.getsize(bam + ".bai"), os.path.getsize(out) / 1e3))
    for f in (bam, bam + ".bai"):
        os.remove(f)
    os.rmdir(tmp)


if __name__ == "__main__":
    sys.exit(main())
