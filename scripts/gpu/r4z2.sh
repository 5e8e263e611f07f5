export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r4z
timeout 300 python -u -m pytest tests/test_gpu_plan.py -m gpu -x -v --timeout 120 --timeout-method=thread > gpurun_out/r4z/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4z/pytest.log
tail -80 gpurun_out/r4z/pytest.log | cut -c1-200
