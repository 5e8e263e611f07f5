export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r4r
run() {
env $2 timeout 1500 python bench.py --config $1 --steps 5 --warmup 2 --no-cpu-baseline --other-configs none --e2e-records 0 --e2e-realistic-records 0 --no-two-files --detail-out gpurun_out/r4r/d.json > gpurun_out/r4r/$1_$3.json 2> gpurun_out/r4r/err.log; tail -1 gpurun_out/r4r/err.log | grep -v amdgpu.ids; python -c "
import json; d=json.load(open('gpurun_out/r4r/$1_$3.json')); print('$1 $3', d['ms_per_step'], d['first_count_ms'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
}
run C5 PC_SMALL_ROWS=0 base
run C5 PC_SMALL_ROWS=1 small
run C5 "PC_SMALL_ROWS=1 PC_SMALL_N=2048" small2k
run C5 "PC_SMALL_ROWS=1 PC_SMALL_N=32768" small32k
run C4 PC_SMALL_ROWS=0 base
