/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_cost_probe scripts/ubench/valu_cost_probe.hip 2>&1 | tail -5
timeout 120 /tmp/valu_cost_probe
