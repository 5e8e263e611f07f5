export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r4s
timeout 1500 python -m pytest tests -m gpu -x -q -k "not fullsize" > gpurun_out/r4s/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4s/pytest.log
tail -15 gpurun_out/r4s/pytest.log | cut -c1-220
timeout 300 python scripts/exp_query_phases.py 2>&1 | grep -v amdgpu.ids
PC_NO_SINGLE=1 timeout 300 python scripts/exp_query_phases.py 2>&1 | grep -v amdgpu.ids
