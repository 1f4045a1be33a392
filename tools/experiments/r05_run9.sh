#!/bin/bash
export PIRGPU_ALLOW_ENV=1
mkdir -p gpurun_out
timeout 600 python tools/rank_budget.py --slots 3 4,8 > gpurun_out/r05_budget_orders_cfg3.log 2>&1
timeout 900 python tools/rank_budget.py --slots 4 8 > gpurun_out/r05_budget_orders_cfg4.log 2>&1
python3 - <<'PY'
import json
for c in (3, 4):
    d = json.load(open("gpurun_out/rank_budget_slots_cfg%d.json" % c))
    for g, r in d.items():
        if isinstance(r, dict):
            print(c, g, {k: v for k, v in r.items() if "order" in k or k in ("S_E_U_queued_together_ms", "E_ms", "S_ms", "U_ms")})
        else:
            print(c, g, r)
PY
