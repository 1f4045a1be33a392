#!/bin/bash
# round 5, first GPU run: the slot-sharded step's parity tests, the scan regression tests (the scan kernel changed its
# addressing), the per-rank budget of the slot-sharded step at cfg 3 and one default bench line
export PIRGPU_ALLOW_ENV=1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_slots.py -x -q > gpurun_out/r05_t_slots.log 2>&1
echo "slots rc=$?" > gpurun_out/r05_run1_rc.txt
timeout 900 python -m pytest tests/test_gpu_mfma_scan.py tests/test_gpu_parity.py tests/test_gpu_distributed.py -x -q > gpurun_out/r05_t_scan.log 2>&1
echo "scan rc=$?" >> gpurun_out/r05_run1_rc.txt
timeout 600 python tools/rank_budget.py --slots 3 1,2,4,8 > gpurun_out/r05_budget_slots_cfg3.log 2>&1
echo "budget rc=$?" >> gpurun_out/r05_run1_rc.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench0.json 2> gpurun_out/r05_bench0.err
echo "bench rc=$?" >> gpurun_out/r05_run1_rc.txt
tail -5 gpurun_out/r05_t_slots.log gpurun_out/r05_t_scan.log gpurun_out/r05_budget_slots_cfg3.log
cat gpurun_out/r05_run1_rc.txt
