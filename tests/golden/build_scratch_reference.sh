set -e
W=${1:-/tmp/oracle}; rm -rf $W && mkdir -p $W && cp -r /root/reference/plastid $W/ && cd $W && chmod -R u+w .
: > plastid/__init__.py                                    # avoid importing BigBed/BigWig/Kent extensions
mkdir -p stubs/pysam stubs/termcolor stubs/Bio stubs/twobitreader
cat > stubs/pysam/__init__.py <<'PY'
class AlignedSegment(object):                              # stub read: what the hot path consumes
    def __init__(self, positions=(), is_reverse=False):
        self.positions = list(positions); self.is_reverse = is_reverse
class AlignmentFile(object): pass
Samfile = AlignmentFile
def get_include(): return []
def get_defines(): return []
__version__ = "0.19.0"
PY
printf 'class tabix_generic_iterator: pass\nclass tabix_file_iterator: pass\nclass asTuple: pass\nclass asGTF: pass\nclass Tabixfile: pass\nTabixFile = Tabixfile\n' > stubs/pysam/libctabix.py
cp stubs/pysam/libctabix.py stubs/pysam/ctabix.py
echo 'def colored(s,*a,**k): return s' > stubs/termcolor/__init__.py
: > stubs/Bio/__init__.py; echo 'class Seq(str): pass' > stubs/Bio/Seq.py
printf 'class SeqRecord(object):\n    def __init__(self,*a,**k): pass\n' > stubs/Bio/SeqRecord.py
echo 'class TwoBitFile(object): pass' > stubs/twobitreader/__init__.py
echo 'class BigWigReader(object): pass' > plastid/readers/bigwig.py      # real one needs the Kent C library
echo 'class BigBedReader(object): pass' > plastid/readers/bigbed.py      # idem (only needed to import bin/psite.py)
# numpy-2 / Cython-3 type aliases only (no arithmetic touched)
sed -i -e 's/^INT    = numpy.int$/INT = numpy.int_/' -e 's/^FLOAT  = numpy.float$/FLOAT = numpy.float64/' -e 's/^LONG   = numpy.long$/LONG = numpy.int64/' \
  -e 's/^ctypedef numpy.int_t    INT_t/ctypedef long INT_t/' -e 's/^ctypedef numpy.float_t  FLOAT_t/ctypedef double FLOAT_t/' \
  -e 's/^ctypedef numpy.double_t DOUBLE_t/ctypedef double DOUBLE_t/' -e 's/^ctypedef numpy.long_t   LONG_t/ctypedef long LONG_t/' \
  -e 's/\blong(\([a-z_]*\))/int(\1)/g' -e 's/\blong(items\[\([12]\)\])/int(items[\1])/' -e 's/\(  *\)long),/\1int),/' plastid/genomics/roitools.pyx
sed -i -e 's/^INT    = np.int$/INT = np.int_/' -e 's/^FLOAT  = np.float$/FLOAT = np.float64/' -e 's/^LONG   = np.long$/LONG = np.int64/' \
  -e 's/^ctypedef np.int_t    INT_t/ctypedef long INT_t/' -e 's/^ctypedef np.float_t  FLOAT_t/ctypedef double FLOAT_t/' \
  -e 's/^ctypedef np.double_t DOUBLE_t/ctypedef double DOUBLE_t/' -e 's/^ctypedef np.long_t   LONG_t/ctypedef long LONG_t/' \
  -e '/^IF PYSAM10:/,/^    from pysam.calignmentfile cimport AlignedSegment/d' \
  -e 's/AlignedSegment read not None/object read/' -e 's/^\( *\)AlignedSegment read$/\1object read/' plastid/genomics/map_factories.pyx
for f in c_common roitools map_factories; do cython -3 -I . plastid/genomics/$f.pyx 2>&1 | grep -A8 "^Error" || true; done
python - <<'PY' >/dev/null 2>&1
from setuptools import setup, Extension; import numpy
setup(name="oracle", script_args=["build_ext","--inplace","-q"], ext_modules=[Extension("plastid.genomics."+n,
      ["plastid/genomics/%s.c"%n], include_dirs=[numpy.get_include()]) for n in ("c_common","roitools","map_factories")])
PY
PYTHONPATH=$W:$W/stubs python -W ignore -c "
from plastid.genomics.genome_array import BAMGenomeArray
from plastid.genomics.roitools import GenomicSegment, SegmentChain
from plastid.genomics.map_factories import *
import pysam, numpy
r=[pysam.AlignedSegment(range(0,L),False) for L in range(25,40)]
ro,c=FivePrimeMapFactory(10)(r,GenomicSegment('mock',0,2000,'+')); assert c[10]==15 and c.sum()==15 and c.dtype==numpy.int64
print('oracle OK at', '$W')"
