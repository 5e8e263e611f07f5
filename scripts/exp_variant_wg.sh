mkdir -p gpurun_out/variants
for flags in "-DPC_HIST_WG=128" "-DPC_HIST_WG=128 -DPC_HIST_U(K)=8"; do
  tag=$(echo "$flags" | tr -d ' =-')
  LIBV=$(python - <<PY
import os
from plastid_amd import build
print(build.build_library(force=True, extra_flags="$flags".split(), out=os.path.abspath("gpurun_out/variants/lib_$tag.so")))
PY
)
  echo "=== $flags -> $LIBV"
  PLASTID_AMD_LIB=$LIBV CONFIG=C4 SCALE=1 TX=1 KNOBS="|PC_SMALL_G=1024;PC_SMALL_N=8192" timeout 900 python scripts/exp_config.py 2>&1 | tail -2
  PLASTID_AMD_LIB=$LIBV CONFIG=C5 SCALE=0.5 TX=1 KNOBS="" timeout 900 python scripts/exp_config.py 2>&1 | tail -1
  PLASTID_AMD_LIB=$LIBV CONFIG=C2 SCALE=1 TX=1 KNOBS="" timeout 900 python scripts/exp_config.py 2>&1 | tail -1
done
