#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
# round 4 profile collection (one gpurun call): tools/collect_profiles.sh r04 + the PMC traffic passes of the scan kernel,
# stamped with the commit and the scan source's hash, + the wire-load kernel trace
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r04 > gpurun_out/collect_r04.log 2>&1
bash tools/pmc_scan_traffic.sh gpurun_out/final/r04_pmc_scan_traffic.json 1679486 3 4 5 > gpurun_out/pmc_traffic_r04.log 2>&1
rm -rf gpurun_out/pmc_scan_cfg*_fetch gpurun_out/pmc_scan_cfg*_write
python3 tools/experiments/r04_wire_load.py 2 20 2>&1 | grep callers > gpurun_out/final/r04_wire_load.txt
python3 tools/experiments/r04_wire_load.py 1 20 2>&1 | grep callers >> gpurun_out/final/r04_wire_load.txt
python3 tools/experiments/r04_wire_load.py 4 20 2>&1 | grep callers >> gpurun_out/final/r04_wire_load.txt
ls -la gpurun_out/final
