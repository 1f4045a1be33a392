#!/bin/bash
# round 6, GPU call 12: upper-level workgroups per query (chunks of children) and the batch pass's share of the chip, re-swept at
# the round's final kernels
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6l; mkdir -p $O
tools/experiments/r06_ab.sh $O 2 3 "--steps 20 --warmup 5" head head:PIRGPU_UPPER_BLOCKS_BATCH=96 head:PIRGPU_UPPER_BLOCKS_BATCH=128 head:PIRGPU_UPPER_BLOCKS_BATCH=192 head:PIRGPU_SCAN_MFMA_WGS_BATCH=112 head:PIRGPU_SCAN_MFMA_WGS_BATCH=144 head:PIRGPU_SCAN_MFMA_WGS_BATCH=160 > $O/summary_cfg3.txt 2>&1
cut -c1-150 $O/summary_cfg3.txt | grep MEAN
