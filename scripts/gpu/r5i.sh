export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r5i
bash scripts/gpu/r5h.sh 2>&1 | grep PASS
timeout 2400 python -u -m pytest tests -m gpu -x -q --timeout 900 --timeout-method=thread > gpurun_out/r5i/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5i/pytest.log
tail -6 gpurun_out/r5i/pytest.log | cut -c1-250
