#!/bin/bash
# rocprofv3 evidence for the bench workload: kernel trace + stats, then PMC passes (each in its own run).
# usage (on the GPU box, from the repo root): bash scripts/profile.sh <tag> [bench args...]
TAG=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 5 --warmup 1 --no-cpu-baseline $@"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/pmc_sq1 -o pmc --output-format csv -- python3 $R/bench.py $ARGS > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAVES -d $OUT/pmc_sq2 -o pmc --output-format csv -- python3 $R/bench.py $ARGS > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc --output-format csv -- python3 $R/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc --output-format csv -- python3 $R/bench.py $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_tcc -o pmc --output-format csv -- python3 $R/bench.py $ARGS > $OUT/pmc_tcc.log 2>&1
cd $R
python3 scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep the merge-back small: drop the bulky raw traces, keep stats + counter CSVs
find $OUT -name "*.db" -delete 2>/dev/null
du -sh $OUT
