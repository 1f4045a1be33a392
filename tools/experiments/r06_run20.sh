#!/bin/bash
# round 6, GPU call 20: the upper level with 8 residues per thread (upper_fused8_kernel, option UPPER_EIGHT): parity, then
# cfg 4 and cfg 3 against the 16-residue form
cd ${GRAFT_REPO_ROOT:-.}
export PIRGPU_ALLOW_ENV=1
O=gpurun_out/r6t; mkdir -p $O
PIRGPU_UPPER_EIGHT=1 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_rings.py tests/test_gpu_ntt_modes.py -x -q -m gpu -k "query or multiply or large or ring or looped or fallback" > $O/tests.log 2>&1; tail -3 $O/tests.log
tools/experiments/r06_ab.sh $O 3 4 "--batch 16 --steps 5 --warmup 2" head head:PIRGPU_UPPER_EIGHT=1 > $O/summary_cfg4.txt 2>&1
cut -c1-230 $O/summary_cfg4.txt
tools/experiments/r06_ab.sh $O 3 3 "--steps 20 --warmup 5" head head:PIRGPU_UPPER_EIGHT=1 > $O/summary_cfg3.txt 2>&1
cut -c1-230 $O/summary_cfg3.txt
