#!/bin/bash
export PIRGPU_ALLOW_ENV=1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/final5b; rm -rf $O; mkdir -p $O
bash tools/pmc_scan_traffic.sh $O/r05_pmc_scan_traffic.json ${1:-unknown} 3 4 5 > $O/pmc_scan.log 2>&1
rm -rf gpurun_out/pmc_scan_cfg*_fetch gpurun_out/pmc_scan_cfg*_write
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_suite.log 2>&1
echo "suite rc=$?" > $O/rc.txt
tail -3 $O/gpu_suite.log; grep "traffic_over_algorithmic" $O/r05_pmc_scan_traffic.json
