export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r4q
timeout 2400 python bench.py --detail-out gpurun_out/r4q/bench_detail.json > gpurun_out/r4q/bench.json 2> gpurun_out/r4q/bench.err; echo "bench rc=$?"
tail -3 gpurun_out/r4q/bench.err; wc -c gpurun_out/r4q/bench.json; cat gpurun_out/r4q/bench.json
