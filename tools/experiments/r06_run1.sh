#!/bin/bash
# round 6, GPU call 1: baseline (commit 0a6251e's library) against the scan without LDS staging / barrier
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6a; mkdir -p $O
PIRGPU_LIB=$PWD/.ab/sdirect/libpirgpu.so timeout 900 python -m pytest tests/test_gpu_mfma_scan.py -x -q -m gpu > $O/test_sdirect.log 2>&1
tail -3 $O/test_sdirect.log
tools/experiments/r06_ab.sh $O 3 3 "--steps 20 --warmup 5" base sdirect base:PIRGPU_SCAN_F64_FOLD=1 sdirect:PIRGPU_SCAN_F64_FOLD=1 > $O/summary_cfg3.txt 2>&1
cat $O/summary_cfg3.txt
tools/experiments/r06_ab.sh $O 1 4 "--batch 16 --steps 5 --warmup 2" base sdirect > $O/summary_cfg4.txt 2>&1
cat $O/summary_cfg4.txt
