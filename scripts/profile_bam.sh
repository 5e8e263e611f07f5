#!/bin/bash
# rocprofv3 evidence for the GPU BAM decoder: kernel trace + stats of scripts/exp_bam_gpu.py (N records as an aligner
# writes them), then one PMC pass with the instruction counters.  The program goes directly after `--`.
# usage (on the GPU box, from the repo root): bash scripts/profile_bam.sh <tag> [records]
TAG=${1:-r05_bam}; N=${2:-2e7}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python3 $R/scripts/exp_bam_gpu.py $N realistic > $OUT/exp_bam_gpu.txt 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 $R/scripts/exp_bam_gpu.py $N realistic > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq2 -o pmc --output-format csv -- python3 $R/scripts/exp_bam_gpu.py $N realistic > $OUT/pmc_sq2.log 2>&1
cd $R
python3 scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
head -40 $OUT/summary.txt
find $OUT -name "*.db" -delete 2>/dev/null
find $OUT -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
du -sh $OUT
