export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r4t
export PLASTID_AMD_LIB=$PWD/build_variants/libpc_batch.so
PC_BAM_DEBUG=1 timeout 1200 python -m pytest tests/test_gpu_bam.py -m gpu -x -q > gpurun_out/r4t/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4t/pytest.log
tail -40 gpurun_out/r4t/pytest.log | cut -c1-250
timeout 600 python scripts/exp_bam_gpu.py 3e6 realistic > gpurun_out/r4t/exp_batch.log 2>&1
PC_BGZF_SERIAL=1 timeout 600 python scripts/exp_bam_gpu.py 3e6 realistic > gpurun_out/r4t/exp_serial.log 2>&1
tail -6 gpurun_out/r4t/exp_batch.log | cut -c1-400
tail -6 gpurun_out/r4t/exp_serial.log | cut -c1-400
