#!/bin/bash
# round 6, GPU call 8: key-switch intermediates in 6 / 7 bytes per residue (cfg 4 / cfg 5) against doubles; D2H probe, extended
cd ${GRAFT_REPO_ROOT:-.}
export PIRGPU_ALLOW_ENV=1
O=gpurun_out/r6h; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_large_rings.py tests/test_gpu_ntt_modes.py tests/test_gpu_parity.py -x -q -m gpu > $O/tests.log 2>&1
tail -3 $O/tests.log
tools/experiments/r06_ab.sh $O 2 4 "--batch 16 --steps 5 --warmup 2" head:PIRGPU_PACK_BYTES=8 head > $O/summary_cfg4.txt 2>&1
cut -c1-200 $O/summary_cfg4.txt
tools/experiments/r06_ab.sh $O 1 5 "--batch 16 --steps 3 --warmup 1" head:PIRGPU_PACK_BYTES=8 head > $O/summary_cfg5.txt 2>&1
cut -c1-200 $O/summary_cfg5.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/d2h -o d2h -- $GRAFT_REPO_ROOT/tools/d2h_probe 8 > $GRAFT_REPO_ROOT/$O/d2h_probe.txt 2>&1
cd $GRAFT_REPO_ROOT
grep variant $O/d2h_probe.txt | tail -6
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r6h/d2h/**/*kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    seq = "".join("M" if "marker" in r["Kernel_Name"] else ("b" if "busy" in r["Kernel_Name"] else ("C" if "copyBuffer" in r["Kernel_Name"] else "?")) for r in rows)
    print("kernel sequence (M marker, b busy, C blit copy):", seq)
PY
rm -rf $O/d2h
