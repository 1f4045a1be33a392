#!/bin/bash
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4t; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/tests_all.log
timeout 400 python tools/soak.py 200 > $O/soak.log 2>&1
tail -3 $O/soak.log > $O/soak.tail
for c in 3 4 5; do
  a=""; [ $c = 3 ] || a="--config $c --batch 16 --steps 10"
  PIRGPU_BENCH_SKIP_WIRE=1 timeout 600 python3 bench.py $a --no-cpu-baseline > $O/cfg$c.json 2> /dev/null
done
python3 - <<'PY' > $O/summary.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/r4t/cfg*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f, round(d["value"],1), d.get("latency_ms_single_query"), d.get("phases_ms_single_query"))
PY
