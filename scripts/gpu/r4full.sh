export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r4full
timeout 2400 python -u -m pytest tests -m gpu -x -q --timeout 900 --timeout-method=thread > gpurun_out/r4full/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4full/pytest.log
tail -15 gpurun_out/r4full/pytest.log | cut -c1-250
timeout 1200 python bench.py > gpurun_out/r4full/bench.json 2> gpurun_out/r4full/bench.err; echo "bench rc=$?"
cp bench_detail.json gpurun_out/r4full/ 2>/dev/null
tail -c 3000 gpurun_out/r4full/bench.json
tail -5 gpurun_out/r4full/bench.err | cut -c1-300
