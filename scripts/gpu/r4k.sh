export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r4k
timeout 300 python scripts/gpu/center_check.py 0.002 > gpurun_out/r4k/check.log 2>&1; tail -2 gpurun_out/r4k/check.log
timeout 1200 python -m pytest tests -m gpu -x -q -k "not fullsize" > gpurun_out/r4k/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4k/pytest.log
tail -3 gpurun_out/r4k/pytest.log
run() {
PLASTID_AMD_LIB=$2 timeout 600 python bench.py --config C3 --steps 10 --warmup 2 --no-cpu-baseline --other-configs none --e2e-records 0 --e2e-realistic-records 0 --no-two-files --detail-out gpurun_out/r4k/d.json > gpurun_out/r4k/c3_$1.json 2> gpurun_out/r4k/c3.err; tail -1 gpurun_out/r4k/c3.err | grep -v amdgpu.ids; python -c "
import json; d=json.load(open('gpurun_out/r4k/c3_$1.json')); print('$1', d['ms_per_step'], d['first_count_ms'], d['roofline']['avg_launch_ms'])"
}
run v0 $PWD/build_variants/libv0.so
run new $PWD/plastid_amd/libplastid_counts.so
run v0b $PWD/build_variants/libv0.so
run newb $PWD/plastid_amd/libplastid_counts.so
