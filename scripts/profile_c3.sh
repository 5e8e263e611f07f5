# rocprofv3 passes for the center kernel (C3 at 20 M reads): kernel trace + SQ counters
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/prof_c3; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
ARGS="--config C3 --scale ${C3SCALE:-0.2} --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -d $OUT/pmc_sq1 -o pmc --output-format csv -- python3 $R/bench.py $ARGS > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU -d $OUT/pmc_sq2 -o pmc --output-format csv -- python3 $R/bench.py $ARGS > $OUT/pmc_sq2.log 2>&1
cd $R; python3 scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1; grep -n "k_center" $OUT/summary.txt
find $OUT -name "*.db" -delete 2>/dev/null
