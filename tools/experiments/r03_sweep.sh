#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
# quick same-box sweep of the batch pipeline's run-time knobs (cfg 3, 64 queries per step)
cd $GRAFT_REPO_ROOT
run() { env $1 python3 bench.py --no-cpu-baseline --steps 60 --latency-runs 5 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['value'],1))"; }
for rep in 1 2; do
  run "PIRGPU_LANES=2"
  run "PIRGPU_LANES=3"
  run "PIRGPU_SCAN_MFMA_WGS_BATCH=96"
  run "PIRGPU_SCAN_MFMA_WGS_BATCH=160"
  run "PIRGPU_UPPER_BLOCKS_BATCH=32"
  run "PIRGPU_UPPER_BLOCKS_BATCH=128"
  run "PIRGPU_FUSE_MAC_NODES=64"
  run "PIRGPU_FUSE_MAC_NODES=256"
done
