export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests -m gpu -x -q -k "not fullsize" > gpurun_out/r4a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4a/pytest.log
tail -5 gpurun_out/r4a/pytest.log
timeout 600 python bench.py --config C3 --steps 10 --warmup 2 --no-cpu-baseline --other-configs none --e2e-records 0 --no-two-files > gpurun_out/r4a/c3.json 2> gpurun_out/r4a/c3.err; echo "bench rc=$?"
python scripts/bench_brief.py gpurun_out/r4a/c3.json 2>/dev/null | head -20
PC_CENTER_DEBUG=1 timeout 600 python bench.py --config C3 --steps 1 --warmup 1 --no-cpu-baseline --other-configs none --e2e-records 0 --no-two-files > gpurun_out/r4a/c3_dbg.json 2> gpurun_out/r4a/c3_dbg.err
grep "^\[center\]" gpurun_out/r4a/c3_dbg.err | tail -20
