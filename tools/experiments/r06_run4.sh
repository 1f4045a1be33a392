#!/bin/bash
# round 6, GPU call 4: c0 in NTT form with component 0 as a launch of its own (PIRGPU_C0_NTT=2) against one kernel for both
# components (=1) and the coefficient-form tree (=0); the D2H engine probe
cd ${GRAFT_REPO_ROOT:-.}
export PIRGPU_ALLOW_ENV=1
O=gpurun_out/r6d; mkdir -p $O
PIRGPU_C0_NTT=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ntt_modes.py -x -q -m gpu > $O/tests.log 2>&1
tail -3 $O/tests.log
tools/experiments/r06_ab.sh $O 3 3 "--steps 20 --warmup 5" head:PIRGPU_C0_NTT=0 head:PIRGPU_C0_NTT=2 head:PIRGPU_C0_NTT=1 > $O/summary_cfg3.txt 2>&1
cat $O/summary_cfg3.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/d2h -o d2h -- $GRAFT_REPO_ROOT/tools/d2h_probe 8 > $GRAFT_REPO_ROOT/$O/d2h_probe.txt 2>&1
cd $GRAFT_REPO_ROOT
cat $O/d2h_probe.txt | grep variant
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r6d/d2h/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    names = [r["Kernel_Name"][:40] for r in rows]
    print("kernel trace:", len(rows), "dispatches;", sum(1 for n in names if "copyBuffer" in n), "blit kernels")
    seq = []
    for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
        n = r["Kernel_Name"]
        seq.append("M" if "marker" in n else ("B" if "busy" in n else ("C" if "copyBuffer" in n else "?")))
    print("".join(seq))
for f in glob.glob("gpurun_out/r6d/d2h/**/*memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print("memory copy trace:", len(rows), "copies")
    for r in rows[:30]:
        print({k: r[k] for k in r if k in ("Direction", "Start_Timestamp", "End_Timestamp", "Source_Agent_Id", "Destination_Agent_Id")})
PY
