#!/bin/bash
export PIRGPU_ALLOW_ENV=1
mkdir -p gpurun_out
timeout 1200 python tools/experiments/r05_cfg5_slots_bisect.py 5 > gpurun_out/r05_bisect5.log 2>&1
echo "rc=$?" >> gpurun_out/r05_bisect5.log
grep "variant\|rc=\|Error" gpurun_out/r05_bisect5.log
