export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r4lat
timeout 300 python scripts/exp_query_phases.py > gpurun_out/r4lat/phases.log 2>&1
tail -5 gpurun_out/r4lat/phases.log
timeout 300 python scripts/exp_query_pyprof.py > gpurun_out/r4lat/pyprof.log 2>&1
tail -45 gpurun_out/r4lat/pyprof.log | cut -c1-180
