#!/bin/bash
# round 6, GPU call 3: c0 in NTT form with sigma_g's permutation from a table; four / three waves per SIMD
cd ${GRAFT_REPO_ROOT:-.}
export PIRGPU_ALLOW_ENV=1
O=gpurun_out/r6c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ntt_modes.py -x -q -m gpu > $O/tests.log 2>&1
tail -3 $O/tests.log
tools/experiments/r06_ab.sh $O 3 3 "--steps 20 --warmup 5" head:PIRGPU_C0_NTT=0 head mc3 noperm:PIRGPU_C0_NTT=0 > $O/summary_cfg3.txt 2>&1
cat $O/summary_cfg3.txt
