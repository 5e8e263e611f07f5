export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r5d
run() {
PLASTID_AMD_LIB=$2 timeout 600 python bench.py --config C3 --steps 40 --warmup 3 --no-cpu-baseline --other-configs none --e2e-records 0 --e2e-realistic-records 0 --no-two-files --detail-out gpurun_out/r5d/d.json > gpurun_out/r5d/c3_$1.json 2> gpurun_out/r5d/c3.err; tail -1 gpurun_out/r5d/c3.err | grep -v amdgpu.ids; python -c "
import json; d=json.load(open('gpurun_out/r5d/c3_$1.json')); print('$1', round(d['ms_per_step'],4), d['first_count_ms'], d['roofline']['avg_launch_ms'])"
}
for r in 1 2 3; do
run base$r $PWD/plastid_amd/libplastid_counts.so
run w8$r $PWD/build_variants/libc_w8.so
run s96$r $PWD/build_variants/libc_s96.so
done
