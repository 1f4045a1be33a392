#!/bin/bash
# round 6, GPU call 21: upper-level workgroups per query below the default (64) with the plain form
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6u; mkdir -p $O
tools/experiments/r06_ab.sh $O 3 3 "--steps 20 --warmup 5" head head:PIRGPU_UPPER_BLOCKS_BATCH=32 head:PIRGPU_UPPER_BLOCKS_BATCH=48 head:PIRGPU_UPPER_BLOCKS_BATCH=80 > $O/summary_cfg3.txt 2>&1
grep MEAN $O/summary_cfg3.txt
