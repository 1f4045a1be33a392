#!/bin/bash
# Encode chunks per thread of upper_mac_kernel at cfg 5: 6 / 12 / 24 -> 128.2 / 129.1 / 128.8 queries/s (three runs each):
# flat beyond 6; 12 ships.  Needs an experiment knob PIRGPU_UPPER_MAC_EG (+ an EG = 24 instantiation) that is not in the tree.
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4eg; mkdir -p $O
for rep in 1 2 3; do
  for e in 12 24 6; do
    PIRGPU_UPPER_MAC_EG=$e timeout 600 python3 bench.py --config 5 --batch 16 --steps 10 --no-cpu-baseline > $O/cfg5_eg${e}_$rep.json 2> /dev/null
  done
done
python3 - <<'PY' > $O/summary.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/r4eg/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f, round(d["value"],1), d.get("latency_ms_single_query"), d.get("phases_ms_single_query",{}).get("upper_ms"))
PY
