#!/bin/bash
# Knob sweep on one config (environment knobs read at pc_create).  usage: bash scripts/exp_knobs.sh C5 "PC_WORK_R=32768" "PC_SMALL_N=4096 PC_SMALL_G=1024" ...
CFG=$1; shift
export PC_SYNTH_CACHE=/tmp/synth
run() {
  env $1 python bench.py --config $CFG --other-configs none --no-cpu-baseline --e2e-records 0 --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$CFG', '$1', round(d['ms_per_step'],4), d['config'].get('kernel_ms'))"
}
run "X=0"
for v in "$@"; do run "$v"; done
