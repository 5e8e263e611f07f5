export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r5c
run() {
PLASTID_AMD_LIB=$2 timeout 600 python bench.py --config C3 --steps 10 --warmup 2 --no-cpu-baseline --other-configs none --e2e-records 0 --e2e-realistic-records 0 --no-two-files --detail-out gpurun_out/r5c/d.json > gpurun_out/r5c/c3_$1.json 2> gpurun_out/r5c/c3.err; tail -1 gpurun_out/r5c/c3.err | grep -v amdgpu.ids; python -c "
import json; d=json.load(open('gpurun_out/r5c/c3_$1.json')); print('$1', d['ms_per_step'], d['first_count_ms'], d['roofline']['avg_launch_ms'])"
}
run base $PWD/plastid_amd/libplastid_counts.so
run w8 $PWD/build_variants/libc_w8.so
run s96 $PWD/build_variants/libc_s96.so
run base2 $PWD/plastid_amd/libplastid_counts.so
run w8b $PWD/build_variants/libc_w8.so
run s96b $PWD/build_variants/libc_s96.so
