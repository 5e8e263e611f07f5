# Build-variant sweep on the sparse C4 workload and on C2 (scratch experiment, GPU box).
# usage: bash scripts/exp_variants_c4.sh "<flags1>" "<flags2>" ...
for flags in "$@"; do
  echo "=== variant: $flags"
  python - <<PY
from plastid_amd import build
build.build_library(force=True, extra_flags="$flags".split())
PY
  SCALES=${SCALES:-0.125,0.5} python scripts/exp_small_n.py 2>&1 | grep "scale\|default"
  python bench.py --no-cpu-baseline --steps 50 --warmup 5 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('C2 ms/step', round(d['ms_per_step'],4), d['config']['kernel_ms'])"
done
