export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r4e
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/center_dpp_probe scripts/ubench/center_dpp_probe.hip 2>&1 | tail -5
timeout 300 /tmp/center_dpp_probe 2>&1 | head -24 > gpurun_out/r4e/probe.log; cat gpurun_out/r4e/probe.log
timeout 300 python scripts/gpu/center_check.py 0.002 > gpurun_out/r4e/check.log 2>&1; tail -12 gpurun_out/r4e/check.log
timeout 1200 python -m pytest tests -m gpu -x -q -k "not fullsize" > gpurun_out/r4e/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4e/pytest.log
tail -5 gpurun_out/r4e/pytest.log
timeout 600 python bench.py --config C3 --steps 10 --warmup 2 --no-cpu-baseline --other-configs none --e2e-records 0 --no-two-files --detail-out gpurun_out/r4e/c3_detail.json > gpurun_out/r4e/c3.json 2> gpurun_out/r4e/c3.err; echo "bench rc=$?"; tail -3 gpurun_out/r4e/c3.err; cat gpurun_out/r4e/c3.json
PC_CENTER_DEBUG=1 timeout 600 python bench.py --config C3 --steps 1 --warmup 1 --no-cpu-baseline --other-configs none --e2e-records 0 --no-two-files --detail-out /tmp/d.json > gpurun_out/r4e/c3_dbg.json 2> gpurun_out/r4e/c3_dbg.err
grep "^\[center\]" gpurun_out/r4e/c3_dbg.err | tail -18
