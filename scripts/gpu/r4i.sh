export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r4i
timeout 300 python scripts/gpu/center_check.py 0.002 > gpurun_out/r4i/check.log 2>&1; tail -2 gpurun_out/r4i/check.log
timeout 1200 python -m pytest tests -m gpu -x -q -k "not fullsize" > gpurun_out/r4i/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4i/pytest.log
tail -3 gpurun_out/r4i/pytest.log
timeout 600 python bench.py --config C3 --steps 10 --warmup 2 --no-cpu-baseline --other-configs none --e2e-records 0 --e2e-realistic-records 0 --no-two-files --detail-out gpurun_out/r4i/c3_detail.json > gpurun_out/r4i/c3.json 2> gpurun_out/r4i/c3.err; echo "bench rc=$?"; tail -1 gpurun_out/r4i/c3.err; python -c "
import json; d=json.load(open('gpurun_out/r4i/c3.json')); print(d['ms_per_step'], d['first_count_ms'], d['roofline']['avg_launch_ms'], d['roofline'].get('replay_steps'))"
