export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r5o
PC_BAM_DEBUG=1 timeout 900 python -u -m pytest tests/test_gpu_bam.py -m gpu -x -q --timeout 300 --timeout-method=thread > gpurun_out/r5o/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5o/pytest.log
tail -4 gpurun_out/r5o/pytest.log | cut -c1-250
for piece in 33554432 67108864; do
for ring in 1 0; do
  if [ $ring = 0 ]; then export PC_BAM_NO_RING=1; else unset PC_BAM_NO_RING; fi
  PC_BAM_PIECE=$piece PC_BAM_TIMING=1 timeout 600 python scripts/exp_bam_gpu.py 2e7 realistic > gpurun_out/r5o/exp_${piece}_$ring.log 2>&1
  echo "== piece $piece ring=$ring"; grep "inflate + crc" gpurun_out/r5o/exp_${piece}_$ring.log | tail -2
done; done
unset PC_BAM_NO_RING
bash scripts/gpu/r5j.sh
