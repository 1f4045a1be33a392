#!/usr/bin/env python3
"""A/B builds of libpirgpu.so: tools/build_variant.py NAME --defs "-DX=1 ..." --tu scan_mfma,ntt12 [--src DIR]

Builds .ab/NAME/libpirgpu.so.  Only the translation units named by --tu are recompiled with the extra definitions
(kernels, scan_mfma, ctx, wire, wire_codec, ntt11..ntt14, or `all`); every other object is taken from the in-tree
build (pir_amd/csrc/*.o), which must be current.  --src: compile the named units from another source tree (e.g. a
`git worktree` of an older commit) -- `--tu all --src DIR` gives that commit's library.  The library is selected at run
time with PIRGPU_LIB=.ab/NAME/libpirgpu.so (pir_amd/capi.py); .ab/ is git-ignored but travels with gpurun.
"""
import argparse
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pir_amd import build as B  # noqa: E402

UNITS = {"kernels": ("kernels.hip", None), "scan_mfma": ("scan_mfma.hip", None), "ctx": ("ctx.hip", None),
         "wire": ("wire.cpp", None), "wire_codec": ("wire_codec.cpp", None)}
for n in B.NTT_LOGNS:
    for pb in B.NTT_PACK_BYTES:
        UNITS["ntt%d" % n if pb == 5 else "ntt%dp%d" % (n, pb)] = (B.NTT_SOURCE, (n, pb))


def obj_name(unit):
    src, deg = UNITS[unit]
    return B.ntt_object(*deg) if deg else src.rsplit(".", 1)[0] + ".o"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--defs", default="")
    ap.add_argument("--tu", default="all")
    ap.add_argument("--src", default=None)
    a = ap.parse_args()
    B.build()  # the in-tree objects the variant borrows
    units = list(UNITS) if a.tu == "all" else [u for u in a.tu.split(",") if u]
    if "ntt" in units:       # every degree and width
        units = [u for u in units if u != "ntt"] + [u for u in UNITS if u.startswith("ntt")]
    out = os.path.join(ROOT, ".ab", a.name)
    os.makedirs(out, exist_ok=True)
    csrc = os.path.join(a.src, "pir_amd", "csrc") if a.src else B.CSRC
    flags = [f for f in B.FLAGS] + a.defs.split()
    jobs, objs = [], []
    for u in UNITS:
        if u in units:
            src, deg = UNITS[u]
            o = os.path.join(out, obj_name(u))
            cmd = [B.HIPCC] + flags + (["-x", "hip"] if src.endswith(".hip") else []) + \
                  (["-DPIRGPU_LOGN=%d" % deg[0], "-DPIRGPU_PACK_BYTES=%d" % deg[1]] if deg else []) + \
                  ["-c", os.path.join(csrc, src), "-o", o]
            jobs.append(cmd)
            objs.append(o)
        else:
            objs.append(os.path.join(B.CSRC, obj_name(u)))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(lambda c: subprocess.run(c, check=True), jobs))
    lib = os.path.join(out, "libpirgpu.so")
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    for o in objs:
        if o.startswith(out):
            os.remove(o)
    print(lib)


if __name__ == "__main__":
    main()
