#!/bin/bash
# round 6, GPU call 16: upper_fused's plain form prefetching the next child's source words (upf1) and this child's first
# selector polynomial as well (upf2) across the transform
cd ${GRAFT_REPO_ROOT:-.}
export PIRGPU_ALLOW_ENV=1
O=gpurun_out/r6p; mkdir -p $O
PIRGPU_LIB=$PWD/.ab/upf2/libpirgpu.so PIRGPU_UPPER_LDS_TW=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_rings.py -x -q -m gpu -k "query or multiply or large or ring" > $O/tests.log 2>&1; tail -2 $O/tests.log
tools/experiments/r06_ab.sh $O 3 4 "--batch 16 --steps 5 --warmup 2" head upf1 upf2 > $O/summary_cfg4.txt 2>&1
cut -c1-130 $O/summary_cfg4.txt
tools/experiments/r06_ab.sh $O 2 3 "--steps 20 --warmup 5" head upf1:PIRGPU_UPPER_LDS_TW=0 upf2:PIRGPU_UPPER_LDS_TW=0 > $O/summary_cfg3.txt 2>&1
grep MEAN $O/summary_cfg3.txt
