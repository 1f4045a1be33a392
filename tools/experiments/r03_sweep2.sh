#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
# same-box sweep of the batch pipeline's launch-shape knobs at the final state (cfg 3, 64 queries per step)
cd $GRAFT_REPO_ROOT
run() { env $1 PIRGPU_BENCH_WIRE_CLIENTS=1 python3 bench.py --no-cpu-baseline --steps 80 --latency-runs 5 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['value'],1))"; }
for rep in 1 2; do
  run "PIRGPU_LANES=2"
  run "PIRGPU_SCAN_MFMA_WGS_BATCH=96"
  run "PIRGPU_SCAN_MFMA_WGS_BATCH=112"
  run "PIRGPU_SCAN_MFMA_WGS_BATCH=144"
  run "PIRGPU_SCAN_MFMA_WGS_BATCH=160"
  run "PIRGPU_UPPER_BLOCKS_BATCH=48"
  run "PIRGPU_UPPER_BLOCKS_BATCH=96"
done
