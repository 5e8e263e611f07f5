export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r5b
timeout 2400 python -u -m pytest tests -m gpu -x -q --timeout 900 --timeout-method=thread > gpurun_out/r5b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5b/pytest.log
tail -15 gpurun_out/r5b/pytest.log | cut -c1-250
timeout 300 python scripts/exp_plan_gpu.py > gpurun_out/r5b/plan.log 2>&1; tail -4 gpurun_out/r5b/plan.log
