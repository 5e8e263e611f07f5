/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/dpp_semantics scripts/ubench/dpp_semantics.hip 2>&1 | tail -5
timeout 60 /tmp/dpp_semantics | tail -22
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/center_dpp_probe scripts/ubench/center_dpp_probe.hip 2>&1 | tail -5
timeout 300 /tmp/center_dpp_probe
