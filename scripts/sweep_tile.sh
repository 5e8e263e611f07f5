for G in 2048 4096; do for R in 8192 16384 32768; do
  echo "=== G=$G R=$R"; PC_WORK_R=$R PC_TILE_G=$G python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['kernel_ms'], d['config']['tiles'])"
done; done
