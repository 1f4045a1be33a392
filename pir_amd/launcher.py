"""Self-launch of a one-process-per-GPU job on ONE node: `python bench.py --gpus N` (no launcher environment) starts
N fresh rank processes itself instead of silently running one.

The parent never touches a GPU (devices are counted from sysfs, not through HIP): every rank is a fresh child
process with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set the way `torch.distributed.run` sets them,
so the same script runs unchanged under either launcher.  Rank 0's stdout is relayed (the single JSON line), every
rank's stderr passes through, and the parent's exit code is non-zero as soon as any rank fails -- the surviving ranks
are then terminated (by PID) instead of being left blocked in a collective.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import List, Optional, Sequence


def launched_by_a_launcher(environ=os.environ) -> bool:
    """True when the rank environment is already there (torch.distributed.run, or our own spawn)."""
    return "RANK" in environ and "WORLD_SIZE" in environ


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _kfd_gpu_nodes(topology: str = "/sys/class/kfd/kfd/topology/nodes") -> Optional[int]:
    """GPU agents the kernel driver exposes (KFD topology nodes with SIMDs), or None if there is no KFD sysfs tree."""
    if not os.path.isdir(topology):
        return None
    gpus = 0
    for node in os.listdir(topology):
        try:
            with open(os.path.join(topology, node, "properties")) as f:
                props = dict(line.split(None, 1) for line in f if " " in line)
        except OSError:
            continue
        if int(props.get("simd_count", "0").strip() or 0) > 0:
            gpus += 1
    return gpus


def visible_gpus(environ=os.environ) -> int:
    """Devices a rank of this job could use, counted WITHOUT a GPU runtime in this process: the KFD topology in sysfs
    narrowed by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES (the runtime applies them in that
    order); where sysfs has no KFD tree, a short-lived child process asks torch, so that the parent still never
    initialises HIP next to its N rank children."""
    have = _kfd_gpu_nodes()
    if have is None:
        try:
            out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                                 capture_output=True, text=True, timeout=300, env=dict(environ))
            return int(out.stdout.strip().splitlines()[-1])
        except Exception:       # noqa: BLE001 -- no torch / no runtime: nothing usable
            return 0
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if var in environ:
            listed = [x for x in environ[var].split(",") if x.strip() != ""]
            # an entry that is not a valid index / UUID ends the list (runtime semantics): count up to the first "-1"
            usable = 0
            for x in listed:
                if x.strip() == "-1":
                    break
                usable += 1
            have = min(have, usable)
    return have


def spawn_ranks(command: Sequence[str], n: int, *, check_devices: bool = True, stdout=None, stderr=None,
                poll_s: float = 0.2, timeout_s: Optional[float] = None, extra_env: Optional[dict] = None) -> int:
    """Runs `command` as ranks 0..n-1 and returns the job's exit code (0 only if every rank exited 0).

    Rank 0's stdout goes to `stdout` (default: this process's), the other ranks' stdout is discarded (they print
    nothing by contract), every stderr goes to `stderr`."""
    if n < 1:
        raise ValueError("need at least one rank")
    if check_devices:
        have = visible_gpus()
        if n > have:
            print("error: --gpus %d requested but only %d GPU(s) are visible to this process (HIP_VISIBLE_DEVICES / "
                  "ROCR_VISIBLE_DEVICES?); refusing to start a partial job" % (n, have), file=stderr or sys.stderr)
            return 2
    port = free_port()
    procs: List[subprocess.Popen] = []
    try:
        for r in range(n):
            env = dict(os.environ)
            env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                        "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                        "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
            if extra_env:
                env.update(extra_env)
            procs.append(subprocess.Popen(list(command), env=env,
                                          stdout=(stdout if r == 0 else subprocess.DEVNULL), stderr=stderr))
        t0 = time.monotonic()
        rc = 0
        while True:
            codes = [p.poll() for p in procs]
            failed = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if failed:
                rc = failed[0][1] if failed[0][1] > 0 else 1
                print("error: rank %d exited with code %s; stopping the other ranks" % failed[0], file=stderr or sys.stderr)
                break
            if all(c == 0 for c in codes):
                return 0
            if timeout_s is not None and time.monotonic() - t0 > timeout_s:
                print("error: the %d-rank job did not finish within %.0f s" % (n, timeout_s), file=stderr or sys.stderr)
                rc = 124
                break
            time.sleep(poll_s)
        return rc
    finally:
        for p in procs:                       # exact PIDs we started, never a pattern
            if p.poll() is None:
                p.terminate()
        deadline = time.monotonic() + 10
        for p in procs:
            try:
                p.wait(max(0.1, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
