export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r4u
export PLASTID_AMD_LIB=$PWD/build_variants/libpc_batch.so
PC_BAM_TIMING=1 timeout 600 python scripts/exp_bam_gpu.py 3e6 realistic > gpurun_out/r4u/exp_batch.log 2>&1
tail -60 gpurun_out/r4u/exp_batch.log | cut -c1-500
