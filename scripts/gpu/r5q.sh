export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r5q
PC_BAM_DEBUG=1 timeout 900 python -u -m pytest tests/test_gpu_bam.py -m gpu -x -q --timeout 300 --timeout-method=thread > gpurun_out/r5q/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5q/pytest.log
tail -25 gpurun_out/r5q/pytest.log | cut -c1-250
PC_BAM_TIMING=1 timeout 600 python scripts/exp_bam_gpu.py 2e7 realistic > gpurun_out/r5q/exp.log 2>&1
grep "^\[bam\]" gpurun_out/r5q/exp.log | tail -7
bash scripts/gpu/r5j.sh
