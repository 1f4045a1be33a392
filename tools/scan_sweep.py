#!/usr/bin/env python3
"""Sweeps the scan kernel's launch geometry on the bench workload (run on the GPU box).
Each configuration runs in a child process because the knobs are read at context creation."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
configs = []
for limb in (1, 0):
    for rows in (2, 4):
        for block in (64, 128, 256):
            configs.append((limb, rows, block, 1))
print("limb rows block nsplit scan_ms GB/s total_ms")
for limb, rows, block, nsplit in configs:
    env = dict(os.environ, PIRGPU_SCAN_MQ_SINGLE="0", PIRGPU_SCAN_ROWS=str(rows), PIRGPU_SCAN_BLOCK=str(block), PIRGPU_SCAN_NSPLIT=str(nsplit),
               PIRGPU_SCAN_LIMB=str(limb))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        print(limb, rows, block, nsplit, "%.4f" % j["roofline"]["kernel_ms"], "%.0f" % j["roofline"]["achieved"],
              "%.3f" % j["ms_per_step"], flush=True)
    except Exception as e:
        print(limb, rows, block, nsplit, "FAILED", r.stderr[-300:], flush=True)
