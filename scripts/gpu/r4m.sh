export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r4m
timeout 900 python scripts/exp_bam_gpu.py 3e6 realistic 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4m/real.log
timeout 900 python scripts/exp_bam_gpu.py 2e7 skeleton 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4m/skel.log
