export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r5u
run() {
timeout 600 python bench.py --config C3 --steps 40 --warmup 3 --no-cpu-baseline --other-configs none --e2e-records 0 --e2e-realistic-records 0 --no-two-files --detail-out gpurun_out/r5u/d.json > gpurun_out/r5u/c3_$1.json 2> gpurun_out/r5u/c3.err; python -c "
import json; d=json.load(open('gpurun_out/r5u/c3_$1.json')); print('$1', round(d['ms_per_step'],4), d['first_count_ms'], d['roofline']['avg_launch_ms'])"
}
for r in 1 2; do
run base$r
PC_CENTER_T2=8 run t2_8_$r
PC_CENTER_T2=16 run t2_16_$r
PC_CENTER_T2=8 PC_CENTER_FLOOR=65536 run t2_8_f64k_$r
done
