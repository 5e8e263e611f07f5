# rocprofv3 kernel trace of one bench config: CONFIG SCALE TX env (e.g. CONFIG=C4 SCALE=0.04 TX=0.5)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/prof_${CONFIG:-C4}; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 $R/bench.py --config ${CONFIG:-C4} --scale ${SCALE:-0.04} --tx-scale ${TX:-0.5} --steps 5 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1
cd $R; cut -c1-60,400-520 $OUT/trace/trace_kernel_stats.csv | head -12
find $OUT -name "*.db" -delete 2>/dev/null
