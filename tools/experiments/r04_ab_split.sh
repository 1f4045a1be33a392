#!/bin/bash
# the split upper level (looped upper_ntt + chunk-sharing upper_mac) against the fused kernel at N = 8192 and N = 4096
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4w; mkdir -p $O
for rep in 1 2; do
  for v in 0 1; do
    PIRGPU_SPLIT_UPPER=$v timeout 600 python3 bench.py --config 4 --batch 16 --steps 10 --no-cpu-baseline > $O/cfg4_split${v}_$rep.json 2> /dev/null
    PIRGPU_SPLIT_UPPER=$v PIRGPU_BENCH_SKIP_WIRE=1 timeout 600 python3 bench.py --no-cpu-baseline > $O/cfg3_split${v}_$rep.json 2> /dev/null
  done
done
PIRGPU_SPLIT_UPPER=1 PIRGPU_SPLIT_UPPER_MB=6144 timeout 600 python3 bench.py --config 4 --batch 16 --steps 10 --no-cpu-baseline > $O/cfg4_split1_6g.json 2> /dev/null
python3 - <<'PY' > $O/summary.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/r4w/c*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f, round(d["value"],1), d.get("latency_ms_single_query"), d.get("phases_ms_single_query"))
PY
