#!/bin/bash
# rocprofv3 evidence for one bench config: kernel trace + stats, then PMC passes (each in its own run,
# never combined with a trace domain).  The program goes directly after `--` (no env / shell hop).
# usage (on the GPU box, from the repo root): bash scripts/profile.sh <tag> [bench args...]
#   e.g. bash scripts/profile.sh r02_C4 --config C4
# The synthetic reads are cached under /tmp between the passes (PC_SYNTH_CACHE).
TAG=${1:-r02}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export PC_SYNTH_CACHE=/tmp/pc_synth_cache
cd /tmp
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --other-configs none --e2e-records 0 --e2e-realistic-records 0 --no-single-query $@"
python3 $R/bench.py $ARGS > $OUT/bench_line.json 2> $OUT/bench.log     # un-profiled line of the same command (fills the cache)
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/pmc_sq1 -o pmc --output-format csv -- python3 $R/bench.py $ARGS > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAVES -d $OUT/pmc_sq2 -o pmc --output-format csv -- python3 $R/bench.py $ARGS > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc --output-format csv -- python3 $R/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc --output-format csv -- python3 $R/bench.py $ARGS > $OUT/pmc_write.log 2>&1
if [ -z "$PROFILE_SKIP_TCC" ]; then
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_tcc -o pmc --output-format csv -- python3 $R/bench.py $ARGS > $OUT/pmc_tcc.log 2>&1
fi
cd $R
python3 scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | head -60
# keep the merge-back small: drop the bulky raw traces, keep stats + counter CSVs
find $OUT -name "*.db" -delete 2>/dev/null
find $OUT -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
du -sh $OUT
