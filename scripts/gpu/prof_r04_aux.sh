# rocprofv3 kernel traces of the BAM decoder and of the GPU plan builder (round 4), and one PMC pass over the inflate kernel
export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_r04_aux
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r04_aux/bam -o trace --output-format csv -- python3 $R/scripts/exp_bam_gpu.py 2e7 realistic > $R/gpurun_out/prof_r04_aux/bam.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r04_aux/plan -o trace --output-format csv -- python3 $R/scripts/exp_plan_gpu.py > $R/gpurun_out/prof_r04_aux/plan.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $R/gpurun_out/prof_r04_aux/bam_pmc -o pmc --output-format csv -- python3 $R/scripts/exp_bam_gpu.py 3e6 realistic > $R/gpurun_out/prof_r04_aux/bam_pmc.log 2>&1
cd $R
find gpurun_out/prof_r04_aux -name "*kernel_trace.csv" -size +8M -delete
find gpurun_out/prof_r04_aux -name "*counter_collection.csv" -size +8M -delete
tail -4 gpurun_out/prof_r04_aux/bam.log | cut -c1-250; tail -4 gpurun_out/prof_r04_aux/plan.log | cut -c1-250
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/prof_r04_aux/bam_pmc/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'bgzf' not in k: continue
        key = (k[:40], r['Counter_Name'])
        acc[key][0] += float(r['Counter_Value']); acc[key][1] += 1
    with open('gpurun_out/prof_r04_aux/bam_pmc_summary.txt', 'w') as fh:
        for (k, c), (v, n) in sorted(acc.items()):
            fh.write("%-42s %-22s mean per launch %.4g (%d launches)\n" % (k, c, v / n, n))
print(open('gpurun_out/prof_r04_aux/bam_pmc_summary.txt').read())
PY
