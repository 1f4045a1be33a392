#!/usr/bin/env python3
"""From a rocprofv3 kernel trace CSV: how much of the communication kernels' time (RCCL device kernels and the copy
kernels of a local exchange) runs UNDER the library's compute kernels -- evidence that the pipelined rows step
overlaps exchange and compute on real streams.  usage: overlap_from_trace.py <kernel_trace.csv>"""
import csv
import sys


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def overlap(u1, u2):
    i = j = 0
    tot = 0
    while i < len(u1) and j < len(u2):
        a, b = max(u1[i][0], u2[j][0]), min(u1[i][1], u2[j][1])
        if b > a:
            tot += b - a
        if u1[i][1] < u2[j][1]:
            i += 1
        else:
            j += 1
    return tot


def main():
    comm, comp = [], []
    names = {}
    for r in csv.DictReader(open(sys.argv[1])):
        n = r["Kernel_Name"]
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        is_comm = "nccl" in n.lower() or "rccl" in n.lower()
        (comm if is_comm else comp).append((a, b))
        if is_comm:
            names[n.split("(")[0][:60]] = names.get(n.split("(")[0][:60], 0) + 1
    uc, up = union(comm), union(comp)
    tc = sum(b - a for a, b in uc)
    tp = sum(b - a for a, b in up)
    ov = overlap(uc, up)
    span = max(b for _, b in comm + comp) - min(a for a, _ in comm + comp)
    print("communication kernels: %d launches, busy %.1f ms (%s)" % (len(comm), tc / 1e6, ", ".join("%s x%d" % kv for kv in names.items())))
    print("compute kernels:       %d launches, busy %.1f ms" % (len(comp), tp / 1e6))
    print("communication time under compute kernels: %.1f ms = %.0f %% of the communication time" % (ov / 1e6, 100.0 * ov / max(tc, 1)))
    print("trace span %.1f ms; compute busy %.0f %% of it" % (span / 1e6, 100.0 * tp / span))


if __name__ == "__main__":
    main()
