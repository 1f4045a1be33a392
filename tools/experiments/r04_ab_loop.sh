#!/bin/bash
# A/B of LOOP_TRANSFORMS (several transforms of one source polynomial per workgroup at the wide expansion levels and the
# split upper level) on one box: cfg 5 (N = 16384, where it is the default), cfg 4 (N = 8192) and cfg 3 (N = 4096).
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4n; mkdir -p $O
for rep in 1 2; do
  for v in 0 1; do
    PIRGPU_LOOP_TRANSFORMS=$v timeout 600 python3 bench.py --config 5 --batch 16 --steps 10 --no-cpu-baseline > $O/cfg5_loop${v}_$rep.json 2> $O/cfg5_loop${v}_$rep.err
  done
done
for v in 0 1; do
  PIRGPU_LOOP_TRANSFORMS=$v timeout 600 python3 bench.py --config 4 --batch 16 --steps 10 --no-cpu-baseline > $O/cfg4_loop${v}.json 2> /dev/null
  PIRGPU_LOOP_TRANSFORMS=$v PIRGPU_BENCH_SKIP_WIRE=1 timeout 600 python3 bench.py --no-cpu-baseline > $O/cfg3_loop${v}.json 2> /dev/null
done
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  export PIRGPU_LOOP_TRANSFORMS=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$v -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --batch 16 --steps 4 --latency-runs 6 --no-cpu-baseline > /dev/null 2>&1
  cp $(find $GRAFT_REPO_ROOT/$O/prof_$v -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/stats_loop$v.csv 2>/dev/null
  rm -rf $GRAFT_REPO_ROOT/$O/prof_$v
done
unset PIRGPU_LOOP_TRANSFORMS
cd $GRAFT_REPO_ROOT
python3 - <<'PY' > $O/summary.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/r4n/c*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f, round(d["value"],1), d.get("latency_ms_single_query"), d.get("phases_ms_single_query"))
PY
