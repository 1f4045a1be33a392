#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
# A/B on one box: top digit of database + selectors as a nibble (default where the moduli allow) vs as a full byte
cd $GRAFT_REPO_ROOT
run() { env $1 python3 bench.py --no-cpu-baseline --steps ${3:-60} --latency-runs ${4:-100} $2 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=j.get('phases_ms_single_query') or {}
print('$1 $2', 'qps', round(j['value'],1), 'latency_ms', j.get('latency_ms_single_query'), 'scan_ms', t.get('scan_ms'), 'frac', round(j['roofline']['frac'],3), 'bytes', j['roofline'].get('algorithmic_bytes_per_launch'))"; }
for rep in 1 2; do
  run "PIRGPU_SCAN_MFMA_TOP4=1" ""
  run "PIRGPU_SCAN_MFMA_TOP4=0" ""
done
run "PIRGPU_SCAN_MFMA_TOP4=1" "--config 4" 6 10
run "PIRGPU_SCAN_MFMA_TOP4=0" "--config 4" 6 10
run "PIRGPU_SCAN_MFMA_TOP4=1" "--config 5" 3 5
run "PIRGPU_SCAN_MFMA_TOP4=0" "--config 5" 3 5
