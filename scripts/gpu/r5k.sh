export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r5k
PC_BAM_DEBUG=1 timeout 900 python -u -m pytest tests/test_gpu_bam.py -m gpu -x -q --timeout 300 --timeout-method=thread > gpurun_out/r5k/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5k/pytest.log
tail -4 gpurun_out/r5k/pytest.log | cut -c1-250
bash scripts/gpu/r5j.sh
