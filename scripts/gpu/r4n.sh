export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r4n
timeout 900 python -m pytest tests/test_gpu_bam.py -m gpu -x -q 2>&1 | tail -2
timeout 900 python scripts/exp_bam_gpu.py 3e6 realistic 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4n/real.log | cut -c1-330
