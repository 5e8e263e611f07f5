#!/bin/bash
# Tile kernel compiled for more waves per SIMD / fewer loads in flight: C2, C4, C5 per variant.
# usage: bash scripts/exp_hist_occupancy.sh "U W" ...   (PC_HIST_U, PC_HIST_WAVES; variants go to /tmp, loaded through PLASTID_AMD_LIB)
export PC_SYNTH_CACHE=/tmp/synth
for v in "$@"; do
  set -- $v
  LIBV=$(python - <<PY
from plastid_amd import build
print(build.build_library(force=True, extra_flags=["-DPC_HIST_U(K)=$1", "-DPC_HIST_WAVES(K)=$2"], out="/tmp/libpc_u$1w$2.so"))
PY
)
  for c in C2 C4 C5; do
    PLASTID_AMD_LIB=$LIBV python bench.py --config $c --other-configs none --no-cpu-baseline --e2e-records 0 --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('U=$1 W=$2', '$c', round(d['ms_per_step'],4), d['config'].get('kernel_ms'))"
  done
done
