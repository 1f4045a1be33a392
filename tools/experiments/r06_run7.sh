#!/bin/bash
# round 6, GPU call 7: the c0-in-NTT-form tree again, with its product loop's loads in flight together; the stall test;
# the wire path's copy engines
cd ${GRAFT_REPO_ROOT:-.}
export PIRGPU_ALLOW_ENV=1
O=gpurun_out/r6g; mkdir -p $O
PIRGPU_C0_NTT=2 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "expansion or query or request" > $O/tests.log 2>&1
tail -2 $O/tests.log
tools/experiments/r06_ab.sh $O 3 3 "--steps 20 --warmup 5" head:PIRGPU_C0_NTT=0 head:PIRGPU_C0_NTT=2 > $O/summary_cfg3.txt 2>&1
cat $O/summary_cfg3.txt | cut -c1-150
timeout 1200 python -m pytest tests/test_gpu_distributed.py -x -q -m gpu -k "stalled or eight" > $O/test_stall.log 2>&1
tail -3 $O/test_stall.log
bash tools/experiments/r06_run6.sh
