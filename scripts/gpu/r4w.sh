export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r4w
export PLASTID_AMD_LIB=$PWD/build_variants/libpc_batch.so
PC_BAM_DEBUG=1 PC_BAM_PIECE=300000 timeout 1200 python -m pytest tests/test_gpu_bam.py -m gpu -x -q > gpurun_out/r4w/pytest_pieces.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4w/pytest_pieces.log
tail -5 gpurun_out/r4w/pytest_pieces.log | cut -c1-250
PC_BAM_DEBUG=1 timeout 1200 python -m pytest tests/test_gpu_bam.py -m gpu -x -q > gpurun_out/r4w/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4w/pytest.log
tail -5 gpurun_out/r4w/pytest.log | cut -c1-250
PC_BAM_TIMING=1 timeout 600 python scripts/exp_bam_gpu.py 3e6 realistic > gpurun_out/r4w/exp.log 2>&1
tail -14 gpurun_out/r4w/exp.log | cut -c1-330
PC_BAM_PIECE=1000000000 timeout 600 python scripts/exp_bam_gpu.py 3e6 realistic > gpurun_out/r4w/exp_onepiece.log 2>&1
tail -3 gpurun_out/r4w/exp_onepiece.log | cut -c1-330
timeout 900 python scripts/exp_bam_gpu.py 2e7 realistic > gpurun_out/r4w/exp_20m.log 2>&1
tail -6 gpurun_out/r4w/exp_20m.log | cut -c1-330
