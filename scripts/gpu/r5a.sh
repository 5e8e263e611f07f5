export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r5a
PC_BAM_TIMING=1 timeout 600 python scripts/exp_bam_gpu.py 2e7 realistic > gpurun_out/r5a/exp.log 2>&1
grep "^\[bam\]" gpurun_out/r5a/exp.log | tail -8
grep "^gpu\|^host" gpurun_out/r5a/exp.log | cut -c1-700
