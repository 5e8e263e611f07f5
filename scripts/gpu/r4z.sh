export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r4z
timeout 600 python -u -m pytest tests/test_gpu_plan.py -m gpu -x -v > gpurun_out/r4z/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4z/pytest.log
tail -40 gpurun_out/r4z/pytest.log | cut -c1-250
timeout 300 python scripts/exp_plan_gpu.py > gpurun_out/r4z/plan_build.log 2>&1
tail -40 gpurun_out/r4z/plan_build.log | cut -c1-200
