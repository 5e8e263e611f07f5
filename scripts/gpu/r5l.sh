export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r5l
timeout 1200 python bench.py > gpurun_out/r5l/bench.json 2> gpurun_out/r5l/bench.err; echo "bench rc=$?"
cp bench_detail.json gpurun_out/r5l/ 2>/dev/null
python3 -c "
import json
d=json.loads(open('gpurun_out/r5l/bench.json').read().strip().splitlines()[-1])
print(len(json.dumps(d,separators=(',',':'))), 'bytes')
print(json.dumps(d['scopes']))
print(d['ms_per_step'], {k:(v['ms_per_step'], v['plan_build_ms']) for k,v in d['configs'].items()}, d['config']['bench_wall_s'])
"
tail -2 gpurun_out/r5l/bench.err | cut -c1-300
