# Build-variant sweep of the histogram kernel on the GPU box (scratch experiment).
# usage: SCRIPT=scripts/exp_sum.py bash scripts/exp_variants.sh "<flags1>" "<flags2>" ...   (each a set of -D flags)
# Every variant is built into gpurun_out/variants/ and loaded through PLASTID_AMD_LIB: the product
# library plastid_amd/libplastid_counts.so is never overwritten by an experiment build.
mkdir -p gpurun_out/variants
i=0
for flags in "$@"; do
  i=$((i+1))
  echo "=== variant: $flags"
  VARIANT_LIB=$(python - <<PY
import os
from plastid_amd import build
print(build.build_library(force=True, extra_flags="$flags".split() or ["-DPC_VARIANT_DEFAULT"], out=os.path.abspath("gpurun_out/variants/libplastid_counts_v$i.so")))
PY
)
  PLASTID_AMD_LIB=$VARIANT_LIB SIGMAS=${SIGMAS:-1.5} RS=${RS:-32768} python ${SCRIPT:-scripts/exp_hist.py} 2>&1 | grep "tiles="
done
