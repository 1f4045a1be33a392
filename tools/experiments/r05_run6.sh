#!/bin/bash
# round 5, sixth GPU run: slots tests (gather with scalar piece lookup, block-major scan order), budgets with the variants
export PIRGPU_ALLOW_ENV=1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_slots.py -x -q > gpurun_out/r05_t_slots3.log 2>&1
echo "slots rc=$?" > gpurun_out/r05_run6_rc.txt
timeout 600 python tools/rank_budget.py --slots 3 8 > gpurun_out/r05_budget_slots_cfg3_v2.log 2>&1
echo "budget3 rc=$?" >> gpurun_out/r05_run6_rc.txt
timeout 900 python tools/rank_budget.py --slots 4 8 > gpurun_out/r05_budget_slots_cfg4_v2.log 2>&1
echo "budget4 rc=$?" >> gpurun_out/r05_run6_rc.txt
cat gpurun_out/r05_run6_rc.txt; tail -2 gpurun_out/r05_t_slots3.log; grep "G=\|plain" gpurun_out/r05_budget_slots_cfg3_v2.log gpurun_out/r05_budget_slots_cfg4_v2.log
