#!/bin/bash
# round 5, second GPU run: whole GPU suite, per-rank budgets (slots and rows) of cfg 3 / 4 / 5, the slots step through a
# real single-rank RCCL process group
export PIRGPU_ALLOW_ENV=1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r05_t_all.log 2>&1
echo "suite rc=$?" > gpurun_out/r05_run2_rc.txt
PIRGPU_FORCE_DIST=1 PIRGPU_EXCHANGE=slots timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_forced_slots.json 2> gpurun_out/r05_forced_slots.err
echo "forced rc=$?" >> gpurun_out/r05_run2_rc.txt
timeout 600 python tools/rank_budget.py 3 1,2,4,8 > gpurun_out/r05_budget_rows_cfg3.log 2>&1
echo "rows3 rc=$?" >> gpurun_out/r05_run2_rc.txt
timeout 900 python tools/rank_budget.py --slots 4 2,4,8 > gpurun_out/r05_budget_slots_cfg4.log 2>&1
echo "slots4 rc=$?" >> gpurun_out/r05_run2_rc.txt
timeout 900 python tools/rank_budget.py 4 1,8 > gpurun_out/r05_budget_rows_cfg4.log 2>&1
echo "rows4 rc=$?" >> gpurun_out/r05_run2_rc.txt
timeout 900 python tools/rank_budget.py --slots 5 2,4,8 > gpurun_out/r05_budget_slots_cfg5.log 2>&1
echo "slots5 rc=$?" >> gpurun_out/r05_run2_rc.txt
timeout 900 python tools/rank_budget.py 5 1,8 > gpurun_out/r05_budget_rows_cfg5.log 2>&1
echo "rows5 rc=$?" >> gpurun_out/r05_run2_rc.txt
cat gpurun_out/r05_run2_rc.txt
