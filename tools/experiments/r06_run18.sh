#!/bin/bash
# round 6, GPU call 18: upper_fused plain form with its transform's twiddles prefetched one pass ahead (226 registers)
cd ${GRAFT_REPO_ROOT:-.}
export PIRGPU_ALLOW_ENV=1
O=gpurun_out/r6r; mkdir -p $O
tools/experiments/r06_ab.sh $O 4 3 "--steps 20 --warmup 5" head twpf > $O/summary_cfg3.txt 2>&1
cut -c1-120 $O/summary_cfg3.txt
tools/experiments/r06_ab.sh $O 2 4 "--batch 16 --steps 5 --warmup 2" head twpf > $O/summary_cfg4.txt 2>&1
grep MEAN $O/summary_cfg4.txt
