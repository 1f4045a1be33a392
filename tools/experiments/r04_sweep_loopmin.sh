#!/bin/bash
# from how many digit sources per launch on the looped digit kernel pays (option loop_min_sources), cfg 3 / 4 / 5
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4lm; mkdir -p $O
for rep in 1 2; do
for m in 1024 512 2048 4096; do
  PIRGPU_LOOP_MIN_SOURCES=$m PIRGPU_BENCH_SKIP_WIRE=1 timeout 600 python3 bench.py --no-cpu-baseline > $O/cfg3_m${m}_$rep.json 2> /dev/null
done
done
for m in 1024 512 2048 4096; do
  PIRGPU_LOOP_MIN_SOURCES=$m timeout 600 python3 bench.py --config 4 --batch 16 --steps 10 --no-cpu-baseline > $O/cfg4_m${m}.json 2> /dev/null
  PIRGPU_LOOP_MIN_SOURCES=$m timeout 600 python3 bench.py --config 5 --batch 16 --steps 10 --no-cpu-baseline > $O/cfg5_m${m}.json 2> /dev/null
done
python3 - <<'PY' > $O/summary.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/r4lm/c*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f, round(d["value"],1), d.get("latency_ms_single_query"))
PY
