export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r4zc
timeout 300 python scripts/exp_query_phases.py > gpurun_out/r4zc/phases.log 2>&1
tail -3 gpurun_out/r4zc/phases.log
PC_NO_ZERO_COPY=1 timeout 300 python scripts/exp_query_phases.py > gpurun_out/r4zc/phases_copy.log 2>&1
tail -3 gpurun_out/r4zc/phases_copy.log
timeout 2400 python -u -m pytest tests -m gpu -x -q --timeout 900 --timeout-method=thread > gpurun_out/r4zc/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4zc/pytest.log
tail -15 gpurun_out/r4zc/pytest.log | cut -c1-250
