#!/bin/bash
# the remaining launch-shape options at the final state: upper-level workgroups per query in batch mode, the node count
# from which the fused mac+combine kernel takes over
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4sm; mkdir -p $O
for rep in 1 2; do
  PIRGPU_BENCH_SKIP_WIRE=1 timeout 600 python3 bench.py --no-cpu-baseline > $O/cfg3_base_$rep.json 2> /dev/null
  for u in 32 48 96; do
    PIRGPU_UPPER_BLOCKS_BATCH=$u PIRGPU_BENCH_SKIP_WIRE=1 timeout 600 python3 bench.py --no-cpu-baseline > $O/cfg3_ub${u}_$rep.json 2> /dev/null
  done
  for f in 32 64 256; do
    PIRGPU_FUSE_MAC_NODES=$f PIRGPU_BENCH_SKIP_WIRE=1 timeout 600 python3 bench.py --no-cpu-baseline > $O/cfg3_fm${f}_$rep.json 2> /dev/null
  done
done
python3 - <<'PY' > $O/summary.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/r4sm/c*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f, round(d["value"],1), d.get("latency_ms_single_query"))
PY
