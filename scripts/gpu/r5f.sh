export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r5f
timeout 2400 python -u -m pytest tests -m gpu -x -q --timeout 900 --timeout-method=thread > gpurun_out/r5f/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5f/pytest.log
tail -8 gpurun_out/r5f/pytest.log | cut -c1-250
timeout 1200 python bench.py > gpurun_out/r5f/bench.json 2> gpurun_out/r5f/bench.err; echo "bench rc=$?"
cp bench_detail.json gpurun_out/r5f/ 2>/dev/null
python3 -c "
import json
d=json.loads(open('gpurun_out/r5f/bench.json').read().strip().splitlines()[-1])
print(len(json.dumps(d,separators=(',',':'))), 'bytes')
print(json.dumps(d['scopes']))
print(d['ms_per_step'], {k:(v['ms_per_step'], v['plan_build_ms']) for k,v in d['configs'].items()}, d['config']['bench_wall_s'])
"
tail -3 gpurun_out/r5f/bench.err | cut -c1-300
