#!/bin/bash
# the head stream (narrow expansion levels of every group) as a HIGH-priority queue, forced on for the device-resident
# loop (head_mode = 2), against the default (head stream only for asynchronously staged wire batches).  Needs an option
# HEAD_PRIORITY that creates the head stream with hipStreamCreateWithPriority -- NOT in the tree (lost).
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4hp; mkdir -p $O
for rep in 1 2 3; do
  PIRGPU_BENCH_SKIP_WIRE=1 timeout 600 python3 bench.py --no-cpu-baseline > $O/cfg3_base_$rep.json 2> /dev/null
  for lv in 3 5 6; do
    PIRGPU_HEAD_MODE=2 PIRGPU_HEAD_LEVELS=$lv PIRGPU_HEAD_PRIORITY=1 PIRGPU_BENCH_SKIP_WIRE=1 timeout 600 python3 bench.py --no-cpu-baseline > $O/cfg3_prio_lv${lv}_$rep.json 2> /dev/null
  done
  PIRGPU_HEAD_MODE=2 PIRGPU_HEAD_LEVELS=5 PIRGPU_HEAD_PRIORITY=0 PIRGPU_BENCH_SKIP_WIRE=1 timeout 600 python3 bench.py --no-cpu-baseline > $O/cfg3_noprio_lv5_$rep.json 2> /dev/null
done
python3 bench.py > $O/bench_default.json 2> /dev/null
PIRGPU_HEAD_PRIORITY=0 python3 bench.py > $O/bench_noprio.json 2> /dev/null
python3 - <<'PY' > $O/summary.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/r4hp/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    w=d.get("wire_multi_client_qps") or {}
    print(f, round(d["value"],1), d.get("latency_ms_single_query"), w.get("value"), (w.get("single_caller") or {}).get("value"))
PY
