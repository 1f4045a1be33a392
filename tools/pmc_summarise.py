#!/usr/bin/env python3
"""Folds the rocprofv3 --pmc passes written by tools/pmc_kernels.sh into one JSON: per kernel the mean of
every counter over its dispatches, plus derived figures for the VALU-issue roofline of the transform kernels:

  valu_insts_per_wave   SQ_INSTS_VALU / SQ_WAVES
  valu_active_frac      SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES       (share of wave lifetime issuing VALU; both in
                                                                     quad-cycles, MI355X_MICROARCH.md)
  wait_frac             SQ_WAIT_ANY / SQ_WAVE_CYCLES               (parked on s_waitcnt / barrier)
  lds_conflict_frac     SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  hbm_bytes             2 * FETCH_SIZE(KB->B) + WRITE_SIZE         (gfx950: FETCH_SIZE counts 128-B requests at 64 B)
"""
import collections
import csv
import glob
import json
import sys


def short(name):
    n = name.split("(")[0]
    n = n.replace("void ", "").replace("pirgpu::", "")
    return n


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/pmc_%s_*/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            acc["%s grid=%s" % (short(r["Kernel_Name"]), r.get("Grid_Size", "?"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, ctrs in sorted(acc.items()):
        m = {c: sum(v) / len(v) for c, v in ctrs.items()}
        n = max(len(v) for v in ctrs.values())
        d = {"dispatches": n, "mean": m}
        if m.get("SQ_WAVES"):
            d["valu_insts_per_wave"] = m.get("SQ_INSTS_VALU", 0) / m["SQ_WAVES"]
        if m.get("SQ_WAVE_CYCLES"):
            d["valu_active_frac"] = m.get("SQ_ACTIVE_INST_VALU", 0) / m["SQ_WAVE_CYCLES"]
            d["wait_frac"] = m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"]
            d["issue_stall_frac"] = m.get("SQ_WAIT_INST_ANY", 0) / m["SQ_WAVE_CYCLES"]
        if m.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_conflict_frac"] = m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"]
        if "FETCH_SIZE" in m or "WRITE_SIZE" in m:
            d["hbm_bytes"] = 2 * m.get("FETCH_SIZE", 0) * 1024 + m.get("WRITE_SIZE", 0) * 1024
        out[k] = d
    json.dump(out, open("gpurun_out/pmc_%s.json" % tag, "w"), indent=1, sort_keys=True)
    for k, d in out.items():
        print(k, {x: (round(y, 4) if isinstance(y, float) else y) for x, y in d.items() if x != "mean"})


if __name__ == "__main__":
    main()
