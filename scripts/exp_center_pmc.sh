#!/bin/bash
# SQ counters of the center kernel on C3 (instructions per class, LDS activity).  usage: bash scripts/exp_center_pmc.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/synth
cd /tmp
ARGS="--config C3 --steps 2 --warmup 1 --no-cpu-baseline --other-configs none --e2e-records 0"
for q in 0; do
  OUT=$R/gpurun_out/prof_center_q$q
  mkdir -p $OUT
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_WAVE_CYCLES -d $OUT/pmc_a -o pmc --output-format csv -- python3 $R/bench.py $ARGS > $OUT/a.log 2>&1
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY -d $OUT/pmc_b -o pmc --output-format csv -- python3 $R/bench.py $ARGS > $OUT/b.log 2>&1
  python3 - <<PY
import csv, glob, collections
for sub in ("pmc_a", "pmc_b"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not f: print("no csv", sub); continue
    d = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if "k_center" in row["Kernel_Name"] and "weigh" not in row["Kernel_Name"] and "order" not in row["Kernel_Name"]:
            d[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("quad $q", sub, {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(d.items())})
PY
  find $OUT -name "*.db" -delete 2>/dev/null
done
