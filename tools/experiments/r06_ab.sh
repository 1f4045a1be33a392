#!/bin/bash
# Same-box, alternating A/B of libpirgpu.so builds (tools/build_variant.py -> .ab/NAME/libpirgpu.so).
# usage: tools/experiments/r06_ab.sh OUTDIR REPS CONFIG "EXTRA BENCH ARGS" NAME [NAME ...]
#   a NAME may carry environment assignments for the library's gated knobs: "base:PIRGPU_X=1,PIRGPU_Y=2"
export PIRGPU_ALLOW_ENV=1 PIRGPU_BENCH_SKIP_WIRE=1 PIRGPU_BENCH_SKIP_SWEEP=1
cd ${GRAFT_REPO_ROOT:-.}
O=$1; R=$2; CFG=$3; EXTRA=$4; shift 4
mkdir -p $O
for rep in $(seq $R); do
  for spec in "$@"; do
    v=${spec%%:*}; envs=""
    if [ "$spec" != "$v" ]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
    tag=$(echo "$spec" | tr ':=,' '___')
    env PIRGPU_LIB=$PWD/.ab/$v/libpirgpu.so $envs timeout 900 python bench.py --config $CFG --no-cpu-baseline $EXTRA \
      > $O/cfg${CFG}_${tag}_$rep.json 2> $O/cfg${CFG}_${tag}_$rep.err
  done
done
python3 - "$O" "$CFG" <<'PY'
import json, glob, sys, collections
O, cfg = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in sorted(glob.glob("%s/cfg%s_*.json" % (O, cfg))):
    tag = f.split("/")[-1][len("cfg%s_" % cfg):].rsplit("_", 1)[0]
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    bl = d["roofline"].get("batch_launch", {})
    acc[tag].append(d["value"])
    print("%-40s qps %8.1f  lat %.4f  scan_ms %.4f  batch_scan mean %s min %s  phases %s" % (
        f.split("/")[-1], d["value"], d.get("latency_ms_single_query") or 0, d["roofline"]["kernel_ms"],
        bl.get("mean_ms"), bl.get("min_ms"), d.get("phases_ms_single_query")))
for t, v in acc.items():
    print("MEAN %-34s %8.1f  (n=%d, min %.1f max %.1f)" % (t, sum(v) / len(v), len(v), min(v), max(v)))
PY
