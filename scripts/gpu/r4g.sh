export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r4g
timeout 300 python scripts/gpu/center_check.py 0.002 > gpurun_out/r4g/check.log 2>&1; tail -4 gpurun_out/r4g/check.log
timeout 1200 python -m pytest tests -m gpu -x -q -k "not fullsize" > gpurun_out/r4g/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4g/pytest.log
tail -3 gpurun_out/r4g/pytest.log
for w in 8 6 4; do
PC_CENTER_WAVES=$w timeout 600 python bench.py --config C3 --steps 10 --warmup 2 --no-cpu-baseline --other-configs none --e2e-records 0 --e2e-realistic-records 0 --no-two-files --detail-out gpurun_out/r4g/c3_detail.json > gpurun_out/r4g/c3_$w.json 2> gpurun_out/r4g/c3.err; echo "bench rc=$?"; tail -2 gpurun_out/r4g/c3.err; python -c "
import json; d=json.load(open('gpurun_out/r4g/c3_$w.json')); print('waves $w', d['ms_per_step'], d['first_count_ms'], d['roofline']['avg_launch_ms'], d['roofline'].get('replay_steps'))"
done
PC_CENTER_DEBUG=1 timeout 600 python bench.py --config C3 --steps 1 --warmup 1 --no-cpu-baseline --other-configs none --e2e-records 0 --e2e-realistic-records 0 --no-two-files --detail-out /tmp/d.json > gpurun_out/r4g/c3_dbg.json 2> gpurun_out/r4g/c3_dbg.err
grep "^\[center\]" gpurun_out/r4g/c3_dbg.err | tail -14
