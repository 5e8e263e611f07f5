export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r4x
export PLASTID_AMD_LIB=$PWD/build_variants/libpc_batch.so
for piece in 50331648 134217728 268435456 2000000000; do
  PC_BAM_PIECE=$piece PC_BAM_TIMING=1 timeout 900 python scripts/exp_bam_gpu.py 2e7 realistic > gpurun_out/r4x/exp_$piece.log 2>&1
  echo "== piece $piece"; grep "inflate + crc\|member walk" gpurun_out/r4x/exp_$piece.log | tail -4; grep "^gpu" gpurun_out/r4x/exp_$piece.log | tail -2 | cut -c1-250
done
