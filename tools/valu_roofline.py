#!/usr/bin/env python3
"""VALU-issue roofline of the transform kernels from the PMC passes of tools/pmc_kernels.sh.

rocprofv3 serialises dispatches while it collects counters, so the kernel-trace durations of the same passes are
stand-alone durations.  Per (kernel, grid):

  valu_issue_frac = SQ_ACTIVE_INST_VALU [quad-cycles] * 4 / (1024 SIMDs * duration * 2.4 GHz)

i.e. the share of the chip's VALU issue cycles (at the nominal clock) the kernel kept busy; the instructions these
kernels issue are almost all 64-bit (v_fma_f64 / v_mul_f64 / v_add_f64 / v_rndne_f64: one wave64 instruction per
4 cycles per SIMD, measured 31.3 T lane-op/s = 0.80 of the 39.3 T/s that rate implies, profiles/r01_ubench_*).
Also: VALU instructions per wave (and per butterfly: an N-point transform is N/2 * log2 N butterflies over N/16
threads), the share of wave lifetime parked on s_waitcnt/barriers, LDS bank-conflict share, HBM bytes.

usage: python tools/valu_roofline.py <tag> [out.json]     (reads gpurun_out/pmc_<tag>.json + pmc_<tag>_valu/)
"""
import collections
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

SIMDS, CLK = 1024, 2.4e9
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the sources every transform kernel of the table is built from, in this order (bench.py: kernel_sources_sha16)
KERNEL_SOURCES = ("ntt_kernels.hip", "ntt_core.h", "kernels.hip", "arith.h")


def kernel_sources_sha16():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "pir_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def head_commit():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=9", "HEAD"], capture_output=True, text=True).stdout.strip() or "unknown"
    except Exception:
        return "unknown"


def short(name):
    return name.split("(")[0].replace("void ", "").replace("pirgpu::", "")


def main():
    tag = sys.argv[1]
    out_path = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/valu_roofline_%s.json" % tag
    pmc = json.load(open("gpurun_out/pmc_%s.json" % tag))
    dur = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/pmc_%s_valu/**/*kernel_trace.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            dur["%s grid=%d" % (short(r["Kernel_Name"]), g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = {}
    for key, d in pmc.items():
        m = d["mean"]
        if key not in dur or "SQ_ACTIVE_INST_VALU" not in m or not m.get("SQ_WAVES"):
            continue
        ns = sum(dur[key]) / len(dur[key])
        logn = {"deg11": 11, "deg12": 12, "deg13": 13, "deg14": 14}.get(key.split("::")[0])
        row = {
            "dispatches": d["dispatches"], "duration_us": ns / 1e3, "waves": m["SQ_WAVES"],
            "valu_insts_per_wave": m["SQ_INSTS_VALU"] / m["SQ_WAVES"],
            "valu_issue_frac": m["SQ_ACTIVE_INST_VALU"] * 4 / (SIMDS * ns * 1e-9 * CLK),
            "wave_wait_frac": d.get("wait_frac"), "wave_issue_stall_frac": d.get("issue_stall_frac"),
            "lds_conflict_frac": d.get("lds_conflict_frac"), "hbm_bytes": d.get("hbm_bytes"),
        }
        if row["hbm_bytes"]:
            row["hbm_GBps"] = row["hbm_bytes"] / ns
        # (the looped ks_digit form -- fourth template argument true -- runs k + 1 transforms per wave: no per-butterfly figure)
        looped = "ks_digit_kernel<" in key and key.split(">")[0].endswith("true") and key.split(">")[0].count(",") == 3
        if logn and not looped and any(t in key for t in ("ntt_batch", "ct_ntt_fwd", "ks_digit", "ks_mac_intt")):
            row["valu_insts_per_butterfly"] = row["valu_insts_per_wave"] / (logn * 8)   # 16 residues/thread: 8 log2 N
        rows[key] = row
    keep = {k: v for k, v in rows.items() if v["duration_us"] >= 20 or "upper_fused" in k or "ks_last" in k}
    out = {"peak": {"bound": "valu issue", "wave64_fp64_instruction_cycles_per_simd": 4, "simds": SIMDS, "clock_hz": CLK,
                    "unit": "fraction of VALU issue cycles at the nominal clock"},
           "source": "rocprofv3 --pmc passes (tools/pmc_kernels.sh %s); stand-alone durations from the same passes" % tag,
           # the transform kernels' sources as profiled (the snapshot on the GPU box): bench.py withholds the table when the
           # sources it runs differ, tools/check_pmc_fresh.py refuses a file whose stamp is not the committed sources
           "kernel_sources_sha16": kernel_sources_sha16(), "kernel_sources": list(KERNEL_SOURCES),
           "commit": os.environ.get("PIRGPU_PROFILED_COMMIT") or head_commit(),
           "kernels": dict(sorted(keep.items(), key=lambda kv: -kv[1]["duration_us"] * kv[1]["dispatches"]))}
    json.dump(out, open(out_path, "w"), indent=1)
    for k, v in out["kernels"].items():
        print("%-58s %8.1f us  valu/wave %7.0f  issue %.2f  wait %.2f  hbm %s" % (
            k[:58], v["duration_us"], v["valu_insts_per_wave"], v["valu_issue_frac"], v["wave_wait_frac"] or 0,
            ("%.0f GB/s" % v["hbm_GBps"]) if v.get("hbm_GBps") else "-"))


if __name__ == "__main__":
    main()
