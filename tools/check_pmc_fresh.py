#!/usr/bin/env python3
"""Refuses a PMC traffic file that describes an older scan kernel: run here (git is available) on a file merged back
from the GPU box, before it is copied into profiles/.

    python3 tools/check_pmc_fresh.py gpurun_out/r04_pmc_scan_traffic.json [--rename]

Exit 0 when the file's `scan_source_sha16` is the sha256 of the committed pir_amd/csrc/scan_mfma.hip AND its `commit`
contains the last commit that touched that source; exit 1 otherwise (with --rename the file is moved to *.stale so that
it cannot be committed by accident).  bench.py applies the same hash test at run time and reports
`roofline.traffic_stale`."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "pir_amd/csrc/scan_mfma.hip"


def git(*args):
    return subprocess.run(["git", "-C", ROOT] + list(args), capture_output=True, text=True)


def main():
    path = sys.argv[1]
    pm = json.load(open(path))
    problems = []
    have = hashlib.sha256(open(os.path.join(ROOT, SRC), "rb").read()).hexdigest()[:16]
    if pm.get("scan_source_sha16") != have:
        problems.append("scan_source_sha16 %s != %s (sha256 of the working tree's %s)" % (pm.get("scan_source_sha16"), have, SRC))
    last = git("log", "-1", "--format=%H", "--", SRC).stdout.strip()
    commit = str(pm.get("commit", ""))
    if last and commit and commit != "unknown":
        if git("merge-base", "--is-ancestor", last, commit).returncode != 0:
            problems.append("commit %s does not contain %s, the last change to %s" % (commit, last[:9], SRC))
    if git("status", "--porcelain", "--", SRC).stdout.strip():
        problems.append("%s has uncommitted changes: commit first, profile that commit" % SRC)
    if problems:
        print("STALE PMC file %s:\n  " % path + "\n  ".join(problems), file=sys.stderr)
        if "--rename" in sys.argv:
            os.rename(path, path + ".stale")
        return 1
    print("%s: fresh (source %s, commit %s)" % (path, have, commit))
    return 0


if __name__ == "__main__":
    sys.exit(main())
