#!/bin/bash
# three alternating same-box runs: LOOP_TRANSFORMS off / on at cfg 3 and cfg 4; and the split upper level taking its
# children in NTT form (PIRGPU_UPPER_SRC_NTT -- that option lost and is no longer in the tree)
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s; mkdir -p $O
for rep in 1 2 3; do
  for v in 0 1; do
    PIRGPU_UPPER_SRC_NTT=$v timeout 600 python3 bench.py --config 5 --batch 16 --steps 10 --no-cpu-baseline > $O/cfg5_srcntt${v}_$rep.json 2> /dev/null
  done
done
for rep in 1 2 3; do
  for v in 0 1; do
    PIRGPU_LOOP_TRANSFORMS=$v PIRGPU_BENCH_SKIP_WIRE=1 timeout 600 python3 bench.py --no-cpu-baseline > $O/cfg3_loop${v}_$rep.json 2> /dev/null
    PIRGPU_LOOP_TRANSFORMS=$v timeout 600 python3 bench.py --config 4 --batch 16 --steps 10 --no-cpu-baseline > $O/cfg4_loop${v}_$rep.json 2> /dev/null
  done
done
python3 - <<'PY' > $O/summary.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/r4s/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f, round(d["value"],1), d.get("latency_ms_single_query"), d.get("phases_ms_single_query"))
PY
