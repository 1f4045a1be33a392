#!/bin/bash
# round 6, GPU call 17: cfg 3, upper_fused: LDS-twiddle form (head) / plain form / plain form with the source prefetch (upf1), 4 reps
cd ${GRAFT_REPO_ROOT:-.}
export PIRGPU_ALLOW_ENV=1
O=gpurun_out/r6q; mkdir -p $O
tools/experiments/r06_ab.sh $O 4 3 "--steps 20 --warmup 5" head head:PIRGPU_UPPER_LDS_TW=0 upf1:PIRGPU_UPPER_LDS_TW=0 upf1 > $O/summary_cfg3.txt 2>&1
cut -c1-150 $O/summary_cfg3.txt
tools/experiments/r06_ab.sh $O 1 2 "--steps 20 --warmup 5" head upf1:PIRGPU_UPPER_LDS_TW=0 > $O/summary_cfg2.txt 2>&1
grep MEAN $O/summary_cfg2.txt
