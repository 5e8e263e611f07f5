# Build-variant sweep of the histogram kernel on the GPU box (scratch experiment).
# usage: SCRIPT=scripts/exp_sum.py bash scripts/exp_variants.sh "<flags1>" "<flags2>" ...   (each a set of -D flags)
for flags in "$@"; do
  echo "=== variant: $flags"
  python - <<PY
from plastid_amd import build
build.build_library(force=True, extra_flags="$flags".split())
PY
  SIGMAS=${SIGMAS:-1.5} RS=${RS:-32768} python ${SCRIPT:-scripts/exp_hist.py} 2>&1 | grep "tiles="
done
