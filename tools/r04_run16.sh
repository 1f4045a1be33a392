#!/bin/bash
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4q; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ntt_modes.py tests/test_gpu_full_size.py -m gpu -x -q -k "split or loop or cfg5 or flavours" 2>&1 | tail -8 > $O/tests.log
for rep in 1 2; do
  timeout 600 python3 bench.py --config 5 --batch 16 --steps 10 --no-cpu-baseline > $O/cfg5_$rep.json 2> $O/cfg5_$rep.err
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --batch 16 --steps 4 --latency-runs 6 --no-cpu-baseline > /dev/null 2>&1
cp $(find $GRAFT_REPO_ROOT/$O/prof -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/stats.csv 2>/dev/null
rm -rf $GRAFT_REPO_ROOT/$O/prof
cd $GRAFT_REPO_ROOT
python3 - <<'PY' > $O/summary.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/r4q/c*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f, round(d["value"],1), d.get("latency_ms_single_query"), d.get("phases_ms_single_query"))
PY
