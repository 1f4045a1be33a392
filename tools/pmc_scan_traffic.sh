#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
# HBM traffic of the single-query scan launch from rocprofv3 PMC counters, for BASELINE configs 3, 4 and 5:
# FETCH_SIZE and WRITE_SIZE in separate passes (they do not fit one pass), kernel trace only.  Run on the GPU box
# from the repo root:
#   bash tools/pmc_scan_traffic.sh <out.json> [commit] [configs...]
# gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half the bytes of a 16 B/lane coalesced
# streaming read -> x2; WRITE_SIZE as reported.
export TMPDIR=/tmp
# only the workload's own launches: the wire legs and the reference sweep of the default line run the same kernel on other
# (tiny) databases -- their full-grid launches with the smallest write volume would be mistaken for the single-query pass
export PIRGPU_BENCH_SKIP_WIRE=1 PIRGPU_BENCH_SKIP_SWEEP=1
OUT=${1:-gpurun_out/pmc_scan_traffic.json}
COMMIT=${2:-unknown}
shift; shift
CFGS=${@:-3 4 5}
for cfg in $CFGS; do
  for c in FETCH_SIZE WRITE_SIZE; do
    d=gpurun_out/pmc_scan_cfg${cfg}_$(echo $c | tr A-Z a-z | cut -d_ -f1)
    rm -rf $d
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o pmc -- \
      python3 bench.py --config $cfg --no-cpu-baseline --batch 8 --workers 8 --steps 1 --warmup 1 --latency-runs 4 > $d.json 2> $d.err
  done
done
python3 - "$OUT" "$COMMIT" $CFGS <<'PY'
import csv, glob, hashlib, json, collections, sys
out_path, commit, cfgs = sys.argv[1], sys.argv[2], sys.argv[3:]
# the scan kernel's source as profiled: bench.py compares it with the source it runs and flags roofline.traffic as stale
# when they differ; tools/check_pmc_fresh.py refuses a file whose stamp is not the committed source
src_sha = hashlib.sha256(open("pir_amd/csrc/scan_mfma.hip", "rb").read()).hexdigest()[:16]
out = {"collection": "bash tools/pmc_scan_traffic.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with "
                     "--kernel-trace over `bench.py --config C --batch 8 --steps 1 --latency-runs 4`; the single-query "
                     "launches are the ones with the smallest WRITE_SIZE",
       "correction": "FETCH_SIZE x2 on gfx950 for 16 B/lane coalesced streaming reads (MI355X_MICROARCH.md, HBM "
                     "section); WRITE_SIZE as reported; both in KB", "commit": commit, "scan_source_sha16": src_sha, "configs": {}}
for cfg in cfgs:
    per = {}
    for name in ("fetch", "write"):
        rows = []
        for f in glob.glob("gpurun_out/pmc_scan_cfg%s_%s/**/*counter_collection.csv" % (cfg, name), recursive=True):
            for r in csv.DictReader(open(f)):
                if "scan_mfma_kernel" in r["Kernel_Name"]:
                    rows.append((r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r.get("Grid_Size", 0) or 0),
                                 float(r["Counter_Value"])))
        per[name] = rows
    if not per["fetch"] or not per["write"]:
        out["configs"]["cfg" + cfg] = {"error": "no scan_mfma_kernel dispatches found"}
        continue
    # single-query launches: the full-chip grid (the batch launches of 8 queries share the chip: smaller grid) with
    # the smallest write volume
    grid = max(g for _, g, _ in per["fetch"])
    fetch = [v for _, g, v in per["fetch"] if g == grid]
    write = [v for _, g, v in per["write"] if g == grid]
    wmin = min(write)
    write1 = [v for v in write if v <= 1.01 * wmin]
    # fetch launches come in the same order as write launches of the other pass: pair by position
    fetch1 = [f for f, w in zip(fetch, write) if w <= 1.01 * wmin] or fetch
    j = json.loads(open("gpurun_out/pmc_scan_cfg%s_fetch.json" % cfg).read().strip().splitlines()[-1])
    alg = j["roofline"]["algorithmic_bytes"]
    rd, wr = 2 * 1024 * sum(fetch1) / len(fetch1), 1024 * sum(write1) / len(write1)
    out["configs"]["cfg" + cfg] = {
        "kernel": per["fetch"][0][0], "grid_threads": grid, "launches": len(fetch1),
        "FETCH_SIZE_KB_mean": sum(fetch1) / len(fetch1), "WRITE_SIZE_KB_mean": sum(write1) / len(write1),
        "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr,
        "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (rd + wr) / alg,
        "workload": j["config"]["workload"], "roofline_kernel": j["roofline"]["kernel"]}
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out, indent=1))
PY
