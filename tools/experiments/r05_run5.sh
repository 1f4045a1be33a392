#!/bin/bash
# round 5, fifth GPU run: slots tests with the gathering inverse NTT, its budget at cfg 3, soak with the slot-sharded step
# mixed in, default bench (single caller with two calls in flight)
export PIRGPU_ALLOW_ENV=1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_slots.py -x -q > gpurun_out/r05_t_slots2.log 2>&1
echo "slots rc=$?" > gpurun_out/r05_run5_rc.txt
timeout 600 python tools/rank_budget.py --slots 3 4,8 > gpurun_out/r05_budget_slots_cfg3_gather.log 2>&1
echo "budget rc=$?" >> gpurun_out/r05_run5_rc.txt
timeout 600 python tools/soak.py 300 > gpurun_out/r05_soak.log 2>&1
echo "soak rc=$?" >> gpurun_out/r05_run5_rc.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench2.json 2> gpurun_out/r05_bench2.err
echo "bench rc=$?" >> gpurun_out/r05_run5_rc.txt
cat gpurun_out/r05_run5_rc.txt; tail -3 gpurun_out/r05_soak.log gpurun_out/r05_t_slots2.log; grep "G=" gpurun_out/r05_budget_slots_cfg3_gather.log
