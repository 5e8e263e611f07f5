export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r5m
PC_BAM_DEBUG=1 timeout 900 python -u -m pytest tests/test_gpu_bam.py -m gpu -x -q --timeout 300 --timeout-method=thread > gpurun_out/r5m/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5m/pytest.log
tail -4 gpurun_out/r5m/pytest.log | cut -c1-250
for piece in 33554432 67108864 134217728; do
for one in 0 1; do
  if [ $one = 1 ]; then export PC_BAM_ONE_STREAM=1; else unset PC_BAM_ONE_STREAM; fi
  PC_BAM_PIECE=$piece PC_BAM_TIMING=1 timeout 600 python scripts/exp_bam_gpu.py 2e7 realistic > gpurun_out/r5m/exp_${piece}_$one.log 2>&1
  echo "== piece $piece one_stream=$one"; grep "inflate + crc" gpurun_out/r5m/exp_${piece}_$one.log | tail -2
done; done
