# Build-variant sweep on the sparse C4 workload and on C2 (scratch experiment, GPU box).
# usage: bash scripts/exp_variants_c4.sh "<flags1>" "<flags2>" ...
# Variants go to gpurun_out/variants/ and are loaded through PLASTID_AMD_LIB (the product library is
# never overwritten by an experiment build).
mkdir -p gpurun_out/variants
i=0
for flags in "$@"; do
  i=$((i+1))
  echo "=== variant: $flags"
  VARIANT_LIB=$(python - <<PY
import os
from plastid_amd import build
print(build.build_library(force=True, extra_flags="$flags".split() or ["-DPC_VARIANT_DEFAULT"], out=os.path.abspath("gpurun_out/variants/libplastid_counts_c4v$i.so")))
PY
)
  export PLASTID_AMD_LIB=$VARIANT_LIB
  SCALES=${SCALES:-0.125,0.5} python scripts/exp_small_n.py 2>&1 | grep "scale\|default"
  python bench.py --no-cpu-baseline --steps 50 --warmup 5 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('C2 ms/step', round(d['ms_per_step'],4), d['config']['kernel_ms'])"
  unset PLASTID_AMD_LIB
done
