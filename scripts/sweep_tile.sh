mkdir -p gpurun_out
for AGG in 0 1; do for G in 1024 2048 4096; do
  echo "=== AGG=$AGG G=$G"; PC_HIST_AGG=$AGG PC_TILE_G=$G python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['kernel_ms'], d['config']['tiles'])"
done; done
