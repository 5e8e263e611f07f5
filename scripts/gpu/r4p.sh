export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r4p
timeout 1200 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_parity.py -m gpu -x -q -k "rccl or batch or twenty" > gpurun_out/r4p/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4p/pytest.log
tail -30 gpurun_out/r4p/pytest.log | cut -c1-220
