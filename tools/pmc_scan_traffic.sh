#!/bin/bash
# HBM traffic of the scan kernel from rocprofv3 PMC counters: FETCH_SIZE and WRITE_SIZE in separate
# passes (they do not fit one pass), kernel trace only.  Run on the GPU box from the repo root:
#   bash tools/pmc_scan_traffic.sh   -> gpurun_out/pmc_fetch/*.csv, gpurun_out/pmc_write/*.csv
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/pmc_$(echo $c | tr A-Z a-z | cut -d_ -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o pmc -- \
    python3 bench.py --no-cpu-baseline --batch 8 --workers 8 --steps 2 --warmup 1 --latency-runs 6 > $d.log 2>&1
done
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for name in ("fetch", "write"):
    files = glob.glob("gpurun_out/pmc_%s/*counter_collection.csv" % name)
    acc = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if "scan_mfma_kernel" in kn or "sel_pack" in kn:
                acc[(kn.split("(")[0][-40:], r["Counter_Name"], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out["%s | %s | grid %s" % k] = {"mean": sum(v) / len(v), "n": len(v), "min": min(v), "max": max(v)}
json.dump(out, open("gpurun_out/pmc_scan_raw.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
