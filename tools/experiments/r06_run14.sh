#!/bin/bash
# round 6, GPU call 14: the multi-client test after the End-status fix; cfg 4 with one digit transform per workgroup
# (LOOP_TRANSFORMS=0: 118 registers, two 512-thread workgroups per CU) against the looped form (137: one per CU)
cd ${GRAFT_REPO_ROOT:-.}
export PIRGPU_ALLOW_ENV=1
O=gpurun_out/r6n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multi_client.py tests/test_cpp_facade.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
tools/experiments/r06_ab.sh $O 3 4 "--batch 16 --steps 5 --warmup 2" head head:PIRGPU_LOOP_TRANSFORMS=0 > $O/summary_cfg4.txt 2>&1
cut -c1-170 $O/summary_cfg4.txt
