#!/usr/bin/env python3
"""Refuses a PMC-derived file that describes older kernels: run here (git is available) on a file merged back from the
GPU box, before it is copied into profiles/.

    python3 tools/check_pmc_fresh.py gpurun_out/r06_pmc_scan_traffic.json [--rename]
    python3 tools/check_pmc_fresh.py gpurun_out/valu_roofline_r06.json [--rename]

Two kinds of file, told apart by their stamp:
  * `scan_source_sha16` (tools/pmc_scan_traffic.sh): sha256 of pir_amd/csrc/scan_mfma.hip;
  * `kernel_sources_sha16` (tools/valu_roofline.py): sha256 over ntt_kernels.hip + ntt_core.h + kernels.hip + arith.h.
Exit 0 when the stamp is the sha256 of the committed source(s) AND the file's `commit` contains the last commit that
touched them; exit 1 otherwise (with --rename the file is moved to *.stale so that it cannot be committed by accident).
bench.py applies the same hash tests at run time: `roofline.traffic_stale`, `roofline_compute.stale` (table withheld)."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCAN_SOURCES = ("scan_mfma.hip",)
KERNEL_SOURCES = ("ntt_kernels.hip", "ntt_core.h", "kernels.hip", "arith.h")


def git(*args):
    return subprocess.run(["git", "-C", ROOT] + list(args), capture_output=True, text=True)


def sha16(files):
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, "pir_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def check(path, rename=False):
    pm = json.load(open(path))
    problems = []
    if "kernel_sources_sha16" in pm:
        field, files = "kernel_sources_sha16", KERNEL_SOURCES
    else:
        field, files = "scan_source_sha16", SCAN_SOURCES
    rel = ["pir_amd/csrc/" + f for f in files]
    have = sha16(files)
    if pm.get(field) != have:
        problems.append("%s %s != %s (sha256 of the working tree's %s)" % (field, pm.get(field), have, " + ".join(rel)))
    last = git("log", "-1", "--format=%H", "--", *rel).stdout.strip()
    commit = str(pm.get("commit", ""))
    if last and commit and commit != "unknown":
        if git("merge-base", "--is-ancestor", last, commit).returncode != 0:
            problems.append("commit %s does not contain %s, the last change to %s" % (commit, last[:9], " / ".join(rel)))
    if git("status", "--porcelain", "--", *rel).stdout.strip():
        problems.append("%s: uncommitted changes: commit first, profile that commit" % " / ".join(rel))
    if problems:
        print("STALE PMC file %s:\n  " % path + "\n  ".join(problems), file=sys.stderr)
        if rename:
            os.rename(path, path + ".stale")
        return 1
    print("%s: fresh (%s %s, commit %s)" % (path, field, have, commit))
    return 0


def main():
    return check(sys.argv[1], "--rename" in sys.argv)


if __name__ == "__main__":
    sys.exit(main())
