#!/bin/bash
# round 5, third GPU run: whole GPU suite again (one test's expectation fixed), cfg 5 budgets (rows G = 8 after the
# split-upper-level fix for row shards, slots G = 4, 8), the reference sweep
export PIRGPU_ALLOW_ENV=1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05_t_all.log 2>&1
echo "suite rc=$?" > gpurun_out/r05_run3_rc.txt
timeout 900 python bench.py --reference-sweep > gpurun_out/r05_reference_sweep.json 2> gpurun_out/r05_reference_sweep.err
echo "sweep rc=$?" >> gpurun_out/r05_run3_rc.txt
timeout 900 python tools/rank_budget.py --slots 5 4,8 > gpurun_out/r05_budget_slots_cfg5.log 2>&1
echo "slots5 rc=$?" >> gpurun_out/r05_run3_rc.txt
timeout 900 python tools/rank_budget.py 5 8 > gpurun_out/r05_budget_rows_cfg5.log 2>&1
echo "rows5 rc=$?" >> gpurun_out/r05_run3_rc.txt
cat gpurun_out/r05_run3_rc.txt
