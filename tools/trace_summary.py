#!/usr/bin/env python3
"""Per (kernel, grid) table from a rocprofv3 --kernel-trace CSV: dispatches, mean/min duration, total.
usage: trace_summary.py <*_kernel_trace.csv> [min_total_us]"""
import collections
import csv
import sys


def short(name):
    return name.split("(")[0].replace("void ", "").replace("pirgpu::", "")


def main():
    rows = collections.defaultdict(list)
    t0, t1 = None, None
    for r in csv.DictReader(open(sys.argv[1])):
        g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        wg = int(r["Workgroup_Size_X"])
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        rows[(short(r["Kernel_Name"]), g // wg, wg)].append(e - s)
    floor = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    tot_all = sum(sum(v) for v in rows.values())
    print("%-44s %9s %5s %6s %10s %10s %10s %6s" % ("kernel", "wgs", "wg", "calls", "mean_us", "min_us", "total_us", "%"))
    for (k, g, wg), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        tot = sum(v) / 1e3
        if tot < floor:
            continue
        print("%-44s %9d %5d %6d %10.1f %10.1f %10.1f %6.2f" % (k[:44], g, wg, len(v), tot / len(v), min(v) / 1e3, tot,
                                                                 100.0 * sum(v) / tot_all))


if __name__ == "__main__":
    main()
