export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
for W in 4096 2048; do
echo "window $W"
PLASTID_AMD_LIB=$PWD/build_variants/libwin$W.so timeout 900 python -m pytest tests/test_gpu_bam.py -m gpu -x -q 2>&1 | tail -1
PLASTID_AMD_LIB=$PWD/build_variants/libwin$W.so timeout 900 python scripts/exp_bam_gpu.py 3e6 realistic 2>&1 | grep "^gpu" | tail -1 | cut -c1-330
done
echo skeleton 8192
timeout 900 python scripts/exp_bam_gpu.py 2e7 skeleton 2>&1 | grep "^gpu\|^host" | tail -2 | cut -c1-330
