#!/bin/bash
# workgroups of the batch-mode scan pass (the share of the chip it takes beside the other lane) at the final state
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4sw; mkdir -p $O
for rep in 1 2; do
for w in 128 96 112 144 160 192; do
  PIRGPU_SCAN_MFMA_WGS_BATCH=$w PIRGPU_BENCH_SKIP_WIRE=1 timeout 600 python3 bench.py --no-cpu-baseline > $O/cfg3_w${w}_$rep.json 2> /dev/null
done
done
python3 - <<'PY' > $O/summary.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/r4sw/c*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f, round(d["value"],1), d['roofline'].get('batch_launch',{}).get('mean_ms'))
PY
