#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O
python tools/r04_wire_load.py 2 20 > $O/wire_load.log 2>&1
python tools/r04_wire_load.py 1 20 >> $O/wire_load.log 2>&1
HSA_ENABLE_SDMA=0 python tools/r04_wire_load.py 2 20 >> $O/wire_load.log 2>&1
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o w -- python3 $GRAFT_REPO_ROOT/tools/r04_wire_load.py 2 12 >> $GRAFT_REPO_ROOT/$O/wire_load.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/wire_kernel_stats.csv
cp $(find $O/prof -name "*memory_copy_stats.csv" | head -1) $O/wire_memcpy_stats.csv 2>/dev/null
python3 tools/trace_summary.py $(find $O/prof -name "*kernel_trace.csv" | head -1) 1000 > $O/wire_trace_summary.txt 2>&1
python3 - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
f = glob.glob(O + "/prof/**/*kernel_trace.csv", recursive=True)[0]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f)))
# union of kernel intervals over the last 60 % of the trace (steady state)
t0, t1 = iv[0][0], iv[-1][1]
lo = t0 + 0.4 * (t1 - t0)
busy, cur_s, cur_e = 0, None, None
for s, e in iv:
    if e < lo: continue
    s = max(s, lo)
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("GPU busy (union of kernel intervals) over the last 60%% of the run: %.3f" % (busy / (t1 - lo)))
mc = glob.glob(O + "/prof/**/*memory_copy_trace.csv", recursive=True)
if mc:
    rows = list(csv.DictReader(open(mc[0])))
    print("memory copies:", len(rows), "columns", list(rows[0].keys()) if rows else None)
    import collections
    agg = collections.defaultdict(lambda: [0, 0, 0])
    for r in rows:
        k = r.get("Direction", "?")
        agg[k][0] += 1; agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); agg[k][2] += int(r.get("Size", 0) or 0) if "Size" in r else 0
    for k, v in agg.items():
        print("  ", k, "count", v[0], "total ms %.2f" % (v[1] / 1e6), "bytes", v[2])
PY
rm -rf $O/prof
bash tools/r04_ab_head.sh
