export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r5n
PC_BAM_DEBUG=1 PC_BAM_PIECE=200000 timeout 900 python -u -m pytest tests/test_gpu_bam.py -m gpu -x -q --timeout 300 --timeout-method=thread > gpurun_out/r5n/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5n/pytest.log
tail -4 gpurun_out/r5n/pytest.log | cut -c1-250
for piece in 8388608 16777216 33554432 67108864; do
for streams in 2 4; do
  PC_BAM_STREAMS=$streams PC_BAM_PIECE=$piece PC_BAM_TIMING=1 timeout 600 python scripts/exp_bam_gpu.py 2e7 realistic > gpurun_out/r5n/exp_${piece}_$streams.log 2>&1
  echo "== piece $piece streams=$streams"; grep "inflate + crc" gpurun_out/r5n/exp_${piece}_$streams.log | tail -2
done; done
