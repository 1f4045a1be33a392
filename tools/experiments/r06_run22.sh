#!/bin/bash
# round 6, GPU call 22: LDS bank-conflict share of the batch-mode scan launch with the result staging read in one 16-byte
# access (spad2) against the default (two 8-byte reads); separate --pmc pass, kernel trace only
cd ${GRAFT_REPO_ROOT:-.}
export PIRGPU_ALLOW_ENV=1 PIRGPU_BENCH_SKIP_WIRE=1 PIRGPU_BENCH_SKIP_SWEEP=1
O=gpurun_out/r6v; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in head spad2; do
  PIRGPU_LIB=$PWD/.ab/$v/libpirgpu.so rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_$v -o p -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --latency-runs 4 > $O/bench_$v.json 2> $O/bench_$v.err
done
python3 - <<'PY'
import csv, glob, collections
for v in ("head", "spad2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/r6v/pmc_%s/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "scan_mfma_kernel" in r["Kernel_Name"]:
                acc[r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for g, c in sorted(acc.items()):
        m = {k: sum(x) / len(x) for k, x in c.items()}
        if m.get("SQ_LDS_IDX_ACTIVE"):
            print(v, "scan_mfma grid", g, "launches", len(c["SQ_LDS_IDX_ACTIVE"]), "lds_conflict_frac %.3f" % (m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]), "LDS instructions per launch %.0f" % m.get("SQ_INSTS_LDS", 0))
PY
rm -rf $O/pmc_head $O/pmc_spad2
