export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r4y
PC_BAM_DEBUG=1 timeout 1200 python -m pytest tests/test_gpu_bam.py -m gpu -x -q > gpurun_out/r4y/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4y/pytest.log
tail -15 gpurun_out/r4y/pytest.log | cut -c1-250
