"""Self-launch of a one-process-per-GPU job on ONE node: `python bench.py --gpus N` (no launcher environment) starts
N fresh rank processes itself instead of silently running one.

The parent never touches a GPU (it only counts devices, which does not initialise HIP): every rank is a fresh child
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


def visible_gpus() -> int:
    """Devices this process could use; counting them does not initialise the GPU runtime."""
    import torch
    return int(torch.cuda.device_count())


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
