#!/bin/bash
# quick SQ-counter passes only. usage: bash scripts/profile_sq.sh <tag> [bench args]
TAG=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline $@"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/pmc_sq1 -o pmc --output-format csv -- python3 $R/bench.py $ARGS > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_WAVES -d $OUT/pmc_sq2 -o pmc --output-format csv -- python3 $R/bench.py $ARGS > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_IFETCH GRBM_GUI_ACTIVE -d $OUT/pmc_sq3 -o pmc --output-format csv -- python3 $R/bench.py $ARGS > $OUT/pmc_sq3.log 2>&1
cd $R
python3 scripts/summarize_profile.py $OUT 2>&1 | grep -E "^==|k_hist|kernel " 
find $OUT -name "*.db" -delete 2>/dev/null
