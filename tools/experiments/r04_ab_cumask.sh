#!/bin/bash
# every lane on its own half of the CUs (hipExtStreamCreateWithCUMask).  Needs an option LANE_CU_MASK at the lanes'
# stream creation -- an experiment, see DESIGN.md section 9.
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4cm; mkdir -p $O
for rep in 1 2; do
  for p in 0 1 2; do
    PIRGPU_LANE_CU_MASK=$p PIRGPU_BENCH_SKIP_WIRE=1 timeout 600 python3 bench.py --no-cpu-baseline > $O/cfg3_cm${p}_$rep.json 2> $O/cfg3_cm${p}_$rep.err
  done
done
for w in 64 96; do
  PIRGPU_LANE_CU_MASK=1 PIRGPU_SCAN_MFMA_WGS_BATCH=$w PIRGPU_BENCH_SKIP_WIRE=1 timeout 600 python3 bench.py --no-cpu-baseline > $O/cfg3_cm1_w${w}.json 2> /dev/null
done
python3 - <<'PY' > $O/summary.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/r4cm/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f, round(d["value"],1), d.get("latency_ms_single_query"), d.get("batch_reply0_equals_single_query_reply"))
PY
tail -3 $O/cfg3_cm1_1.err >> $O/summary.txt
