export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r4l
PC_BAM_DEBUG=1 timeout 900 python -m pytest tests/test_gpu_bam.py -m gpu -x -q > gpurun_out/r4l/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4l/pytest.log
tail -40 gpurun_out/r4l/pytest.log | cut -c1-200
