export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r5p
PC_BAM_DEBUG=1 timeout 900 python -u -m pytest tests/test_gpu_bam.py -m gpu -x -q --timeout 300 --timeout-method=thread > gpurun_out/r5p/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5p/pytest.log
tail -4 gpurun_out/r5p/pytest.log | cut -c1-250
for r in 1 2; do
for v in new in1k; do
  if [ $v = in1k ]; then export PLASTID_AMD_LIB=$PWD/build_variants/libin1k.so; else unset PLASTID_AMD_LIB; fi
  PC_BAM_TIMING=1 timeout 600 python scripts/exp_bam_gpu.py 2e7 realistic > gpurun_out/r5p/exp_${v}_$r.log 2>&1
  echo "== $v $r"; grep "inflate + crc" gpurun_out/r5p/exp_${v}_$r.log | tail -2
done; done
