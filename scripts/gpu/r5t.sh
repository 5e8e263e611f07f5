export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r5t
run() {
timeout 600 python bench.py --config C3 --steps 30 --warmup 3 --no-cpu-baseline --other-configs none --e2e-records 0 --e2e-realistic-records 0 --no-two-files --detail-out gpurun_out/r5t/d.json > gpurun_out/r5t/c3_$1.json 2> gpurun_out/r5t/c3.err; python -c "
import json; d=json.load(open('gpurun_out/r5t/c3_$1.json')); print('$1', round(d['ms_per_step'],4), d['first_count_ms'], d['roofline']['avg_launch_ms'])"
}
run base1
PC_CENTER_FLOOR=16384 run floor16k
PC_CENTER_FLOOR=65536 run floor64k
PC_CENTER_T1=16 run t1_16
PC_CENTER_T2=2 run t2_2
PC_CENTER_T2=8 run t2_8
run base2
