export PC_SYNTH_CACHE=/tmp/synth
for c in C2 C4 C5; do
python bench.py --config $c --other-configs none --no-cpu-baseline --e2e-records 0 --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$c', round(d['ms_per_step'],4), d['config'].get('kernel_ms'))"
done
