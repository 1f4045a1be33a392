#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
# single-query latency against the upper level's workgroup target (and a few other launch-shape knobs)
cd $GRAFT_REPO_ROOT
run() { env $1 python3 bench.py --no-cpu-baseline --steps 10 --latency-runs 300 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=j.get('phases_ms_single_query') or {}
print('$1', 'latency_ms', j.get('latency_ms_single_query'), {k: round(v,4) for k,v in t.items()})"; }
for v in 512 256 768 1024 1536 2592; do run "PIRGPU_UPPER_BLOCKS=$v"; done
run "PIRGPU_FUSE_MAC_NODES=64"
run "PIRGPU_FUSE_MAC_NODES=256"
run "PIRGPU_FUSE_MAC_NODES=512"
