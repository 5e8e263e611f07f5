export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r4v
R=$GRAFT_REPO_ROOT
for v in batch e1 e2 w2k; do
  PLASTID_AMD_LIB=$R/build_variants/libpc_$v.so PC_BAM_TIMING=1 timeout 300 python scripts/exp_bam_gpu.py 3e6 realistic > gpurun_out/r4v/exp_$v.log 2>&1
  echo "== $v"; grep "inflate + crc" gpurun_out/r4v/exp_$v.log | tail -3; grep "^gpu" gpurun_out/r4v/exp_$v.log | tail -1 | cut -c1-400
done
export PLASTID_AMD_LIB=$R/build_variants/libpc_batch.so
cd /tmp
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  n=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass -d $R/gpurun_out/r4v/pmc_$n -o pmc --output-format csv -- python3 $R/scripts/exp_bam_gpu.py 3e6 realistic > $R/gpurun_out/r4v/pmc_$n.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/r4v/pmc_*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'bgzf' not in k and 'bam' not in k: continue
        key = (k[:40], r['Counter_Name'])
        acc[key][0] += float(r['Counter_Value']); acc[key][1] += 1
    disp = collections.defaultdict(set)
    for (k, c), (v, n) in sorted(acc.items()):
        print("%-42s %-22s sum %.4g over %d rows" % (k, c, v, n))
PY
find gpurun_out/r4v -name "*.csv" -size +5M -delete
