#!/bin/bash
# lanes as HIP streams of different / equal explicit priority.  Needs an option LANE_PRIORITY (1 = lane 0 highest / lane 1
# lowest, 2 = all highest) at the lanes' hipStreamCreate -- NOT in the tree (lost).
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4lp; mkdir -p $O
for rep in 1 2 3; do
  for p in 0 1 2; do
    PIRGPU_LANE_PRIORITY=$p PIRGPU_BENCH_SKIP_WIRE=1 timeout 600 python3 bench.py --no-cpu-baseline > $O/cfg3_lp${p}_$rep.json 2> /dev/null
  done
done
python3 - <<'PY' > $O/summary.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/r4lp/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f, round(d["value"],1), d.get("latency_ms_single_query"))
PY
