# quick look at the other BASELINE configs at reduced size (parity gate included in bench.py)
for cfg in "C3 0.2 1.0" "C4 0.04 0.5" "C5 0.02 0.5" "C2 1.0 1.0"; do
  set -- $cfg
  echo "=== $1 scale=$2 tx=$3"
  python bench.py --config $1 --scale $2 --tx-scale $3 --steps 5 --warmup 1 --cpu-budget 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('reads/s %.3g  ms/step %.3f  records %d  chains %d  pos %d' % (d['value'], d['ms_per_step'], d['config']['records_per_gpu'], d['config']['chains'], d['config']['output_positions_per_gpu']))
print(d['config']['kernel_ms'], 'roofline frac %.3f' % d['roofline']['frac'], 'cpu %.3g' % (d['cpu_baseline']['value'] if d['cpu_baseline'] else 0), d['config']['host_generate_s'], d['config']['host_stage_s'], d['config']['plan_build_ms_once_per_annotation'])
"
done
