#!/bin/bash
# round 6, GPU call 15: upper_fused's plain form with two exchange buffers in turn (one workgroup barrier per child instead of two)
cd ${GRAFT_REPO_ROOT:-.}
export PIRGPU_ALLOW_ENV=1
O=gpurun_out/r6o; mkdir -p $O
PIRGPU_LIB=$PWD/.ab/udb/libpirgpu.so PIRGPU_UPPER_LDS_TW=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_rings.py -x -q -m gpu -k "query or multiply or large or ring" > $O/tests.log 2>&1; tail -2 $O/tests.log
tools/experiments/r06_ab.sh $O 3 4 "--batch 16 --steps 5 --warmup 2" head udb > $O/summary_cfg4.txt 2>&1
cut -c1-170 $O/summary_cfg4.txt | grep -E "MEAN|phases" | cut -c1-120
tools/experiments/r06_ab.sh $O 3 3 "--steps 20 --warmup 5" head head:PIRGPU_UPPER_LDS_TW=0 udb:PIRGPU_UPPER_LDS_TW=0 > $O/summary_cfg3.txt 2>&1
grep MEAN $O/summary_cfg3.txt
