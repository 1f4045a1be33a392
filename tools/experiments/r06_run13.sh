#!/bin/bash
# round 6, GPU call 13: cfg 4 with the looped digit kernel held at 128 registers (36 bytes of scratch; two workgroups per CU)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6m; mkdir -p $O
tools/experiments/r06_ab.sh $O 3 4 "--batch 16 --steps 5 --warmup 2" head dig128 > $O/summary_cfg4.txt 2>&1
cut -c1-170 $O/summary_cfg4.txt
