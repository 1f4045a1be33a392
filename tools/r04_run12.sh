#!/bin/bash
# after the collection: wire tests with the two-part reply download, then the bench lines again so that they carry
# the traffic of the PMC file just committed (roofline.traffic_stale false)
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4m; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -k "wire or multi_client or client" 2>&1 | tail -6 > $O/tests_wire.log
for i in 1 2 3; do python3 bench.py > $O/bench_$i.json 2> $O/bench_$i.err; done
for c in 2 4 5; do python3 bench.py --config $c --batch 16 --steps 10 --no-cpu-baseline > $O/bench_cfg${c}_reference.json 2> $O/cfg$c.err; done
python3 - <<'PY' > $O/summary.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/r4m/bench_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    r=d.get("roofline",{}); w=d.get("wire_process_request_ms",{})
    print(f, d["value"], r.get("frac"), r.get("traffic_stale"), w, d.get("wire_multi_client_qps"), d.get("multi_client_qps"))
PY
