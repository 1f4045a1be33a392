#!/bin/bash
# round 6, GPU call 6: the wire path's copies under the tracer -- blit kernels or SDMA?
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6f${1:-}; mkdir -p $O; export WIRE_O=$O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/wire -o w -- python3 tools/experiments/r06_wire_copy_trace.py 10 > $O/wire_trace.txt 2> $O/wire_trace.err
grep WINDOW $O/wire_trace.txt
python3 - <<'PY'
import csv, glob, collections
import os
O = os.environ["WIRE_O"]
win = [l.split() for l in open(O + "/wire_trace.txt") if l.startswith("WINDOW")][0]
t0, t1 = int(win[1]), int(win[2])
for f in glob.glob(O + "/wire/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    ts = [int(r["Start_Timestamp"]) for r in rows]
    print("kernel trace: %d dispatches, timestamps %d .. %d; window %d .. %d" % (len(rows), min(ts), max(ts), t0, t1))
    inw = [r for r in rows if t0 <= int(r["Start_Timestamp"]) <= t1]
    blit = collections.Counter()
    for r in inw:
        if "copyBuffer" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]:
            g = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
            blit[(r["Kernel_Name"][:40], g, round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, -1))] += 1
    print("inside the timed calls: %d dispatches, blit / runtime kernels: %s" % (len(inw), dict(blit)))
    allb = collections.Counter((r["Kernel_Name"][:40], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) for r in rows if "rocclr" in r["Kernel_Name"])
    print("whole run, runtime kernels:", dict(allb))
for f in glob.glob(O + "/wire/**/*memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    inw = [r for r in rows if t0 <= int(r["Start_Timestamp"]) <= t1]
    c = collections.Counter()
    for r in inw:
        c[(r["Direction"], round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, -1))] += 1
    print("memory copies inside the timed calls: %d of %d; by (direction, ~us): %s" % (len(inw), len(rows), sorted(c.items(), key=lambda kv: -kv[1])[:12]))
PY
rm -rf $O/wire
