export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r4b
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/center_dpp_probe scripts/ubench/center_dpp_probe.hip 2>&1 | tail -5
timeout 300 /tmp/center_dpp_probe > gpurun_out/r4b/probe.log 2>&1; cat gpurun_out/r4b/probe.log
timeout 300 python scripts/gpu/center_check.py 0.002 > gpurun_out/r4b/check.log 2>&1; tail -40 gpurun_out/r4b/check.log
