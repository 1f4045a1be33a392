#!/bin/bash
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4r; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ntt_modes.py tests/test_gpu_full_size.py tests/test_gpu_wire_extras.py tests/test_gpu_multi_client.py -m gpu -x -q 2>&1 | tail -8 > $O/tests.log
for rep in 1 2; do
  timeout 600 python3 bench.py --config 5 --batch 16 --steps 10 --no-cpu-baseline > $O/cfg5_$rep.json 2> $O/cfg5_$rep.err
done
python3 bench.py > $O/bench_1.json 2> $O/bench_1.err
PIRGPU_WIRE_SPLIT=0 python3 bench.py > $O/bench_nosplit.json 2> $O/bench_nosplit.err
python3 - <<'PY' > $O/summary.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/r4r/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    w=d.get("wire_multi_client_qps") or {}
    print(f, round(d["value"],1), d.get("latency_ms_single_query"), d.get("phases_ms_single_query"), w.get("value"), (w.get("single_caller") or {}).get("value"), (d.get("wire_process_request_ms") or {}).get("repeat"))
PY
