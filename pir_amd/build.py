"""Builds libpirgpu.so (hand-written gfx950 HIP kernels + C ABI) in-tree with hipcc."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpirgpu.so")
SOURCES = ["kernels.hip", "scan_mfma.hip", "ctx.hip", "wire.cpp", "wire_codec.cpp"]
NTT_SOURCE = "ntt_kernels.hip"      # compiled once per ring degree (-DPIRGPU_LOGN)
NTT_LOGNS = [11, 12, 13, 14]
NTT_PACK_BYTES = [5, 6, 7]          # ... and once per width of the packed key-switch intermediates (-DPIRGPU_PACK_BYTES)
HEADERS = ["device_params.h", "env_gate.h", "kernels.h", "host_math.h", "wire.h", "wire_codec.h", "arith.h", "ntt_core.h", NTT_SOURCE, os.path.join("..", "..", "include", "pirgpu.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + \
    os.environ.get("PIRGPU_BUILD_DEFS", "").split()      # A/B builds of compile-time choices (tools/experiments/r04_ab_ept.sh)


def ntt_object(logn: int, pb: int) -> str:
    return "ntt_kernels_%d.o" % logn if pb == 5 else "ntt_kernels_%d_p%d.o" % (logn, pb)


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS + [os.path.basename(__file__)]:
        p = os.path.join(CSRC, f) if f != os.path.basename(__file__) else os.path.join(HERE, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    jobs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.rsplit(".", 1)[0] + ".o")
        jobs.append(([HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".hip") else []) +
                     ["-c", os.path.join(CSRC, src), "-o", obj], obj))
    for logn in NTT_LOGNS:
        for pb in NTT_PACK_BYTES:
            obj = os.path.join(CSRC, ntt_object(logn, pb))
            jobs.append(([HIPCC] + FLAGS + ["-x", "hip", "-DPIRGPU_LOGN=%d" % logn, "-DPIRGPU_PACK_BYTES=%d" % pb, "-c",
                          os.path.join(CSRC, NTT_SOURCE), "-o", obj], obj))

    def run(job):
        if verbose:
            print(" ".join(job[0]), file=sys.stderr)
        subprocess.run(job[0], check=True)
        return job[1]

    jobs.sort(key=lambda j: 0 if "ntt_kernels" in j[1] else 1)      # the long compilations first
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(run, jobs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


CLIENT_LIB = os.path.join(HERE, "libpirclient.so")
CLIENT_SOURCES = ["client.cpp", "wire_codec.cpp"]
CLIENT_HEADERS = ["host_math.h", "wire_codec.h", os.path.join("..", "..", "include", "pirgpu.h"),
                  os.path.join("..", "..", "include", "pirclient.h")]
CXX = os.environ.get("CXX", "g++")


def build_client(force: bool = False, verbose: bool = False) -> str:
    """libpirclient.so (include/pirclient.h): host-only C++, no HIP -- the CPU PIR client."""
    deps = [os.path.join(CSRC, f) for f in CLIENT_SOURCES + CLIENT_HEADERS] + [os.path.abspath(__file__)]
    if not force and os.path.exists(CLIENT_LIB) and all(
            os.path.getmtime(d) <= os.path.getmtime(CLIENT_LIB) for d in deps if os.path.exists(d)):
        return CLIENT_LIB
    cmd = [CXX, "-O2", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-shared"] + \
          [os.path.join(CSRC, s) for s in CLIENT_SOURCES] + ["-o", CLIENT_LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return CLIENT_LIB


if __name__ == "__main__":
    print(build_client(force="--force" in sys.argv, verbose=True))
    print(build(force="--force" in sys.argv, verbose=True))
