# usage: bash scripts/gpu/prof_r04.sh <config>   (on the GPU box, from the repo root)
CFG=$1
export TMPDIR=/tmp
bash scripts/profile.sh r04_$CFG --config $CFG --no-two-files > gpurun_out/prof_r04_$CFG.log 2>&1
tail -30 gpurun_out/prof_r04_$CFG.log
if [ "$CFG" = "C2" ]; then
  # the BGZF inflate / BAM decode kernels on the realistic sample (3 M records, 119 bytes each)
  cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04_bam/trace -o trace --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/exp_bam_gpu.py 3e6 realistic > $GRAFT_REPO_ROOT/gpurun_out/prof_r04_bam.log 2>&1
  cd $GRAFT_REPO_ROOT; tail -5 gpurun_out/prof_r04_bam.log | cut -c1-300
  find gpurun_out/prof_r04_bam -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
fi
