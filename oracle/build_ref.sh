#!/bin/bash
# Builds the reference-held htslib (vendored under /root/reference/kent/src/htslib, version 1.3)
# from the sources WHERE THEY LIE, with plain gcc + the system zlib, into oracle/_ref/ (git-ignored;
# never committed, never linked into the product), together with the fixture generator
# tests/golden/hts_golden.c.  Only the build container has /root/reference; on the GPU box this
# script is a no-op and the committed fixtures (tests/golden/hts_fixture.npz) are used.
#
#   bash oracle/build_ref.sh            -> oracle/_ref/libhts_ref.a, oracle/_ref/hts_golden
#
# plastid's hot path itself is Cython + pysam (not C sources that compile on their own), so there
# is no reference binary for the counting path -- see DESIGN.md section 2; what this pins is the
# CIGAR -> positions / fetch / BAM + BAI layer that pysam delegates to htslib.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
H=${HTSLIB_SRC:-/root/reference/kent/src/htslib}
OUT=$HERE/_ref
if [ ! -f "$H/sam.c" ]; then
  echo "build_ref.sh: $H not present (not the build container): nothing to do"
  exit 0
fi
mkdir -p "$OUT/obj"
OBJS=""
for f in kfunc knetfile kstring bgzf faidx hfile hfile_net hts md5 regidx sam synced_bcf_reader vcf_sweep tbx vcf vcfutils \
         cram/cram_codecs cram/cram_decode cram/cram_encode cram/cram_external cram/cram_index cram/cram_io \
         cram/cram_samtools cram/cram_stats cram/files cram/mFILE cram/open_trace_file cram/pooled_alloc \
         cram/rANS_static cram/sam_header cram/string_alloc cram/thread_pool cram/vlen cram/zfio; do
  o="$OUT/obj/$(echo $f | tr / _).o"
  if [ ! -f "$o" ] || [ "$H/$f.c" -nt "$o" ]; then
    gcc -O2 -w -I"$H" -c "$H/$f.c" -o "$o"
  fi
  OBJS="$OBJS $o"
done
rm -f "$OUT/libhts_ref.a"
ar rc "$OUT/libhts_ref.a" $OBJS
gcc -O2 -Wall -I"$H" "$HERE/../tests/golden/hts_golden.c" "$OUT/libhts_ref.a" -lz -lm -lpthread -o "$OUT/hts_golden"
echo "built $OUT/hts_golden against htslib $(sed -n 's/#define HTS_VERSION "\(.*\)"/\1/p' "$H/version.h")"
