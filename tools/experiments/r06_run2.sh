#!/bin/bash
# round 6, GPU call 2: c0 of the expansion tree in NTT form (runtime knob PIRGPU_C0_NTT), ks_mac_combine at four / three
# waves per SIMD, scan staging with one 16-byte LDS read
cd ${GRAFT_REPO_ROOT:-.}
export PIRGPU_ALLOW_ENV=1
O=gpurun_out/r6b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ntt_modes.py tests/test_gpu_mfma_scan.py tests/test_gpu_slots.py -x -q -m gpu > $O/tests.log 2>&1
tail -5 $O/tests.log
tools/experiments/r06_ab.sh $O 3 3 "--steps 20 --warmup 5" base head:PIRGPU_C0_NTT=0 head mc3 spad2 > $O/summary_cfg3.txt 2>&1
cat $O/summary_cfg3.txt
tools/experiments/r06_ab.sh $O 1 2 "--steps 20 --warmup 5" base head > $O/summary_cfg2.txt 2>&1
cat $O/summary_cfg2.txt
