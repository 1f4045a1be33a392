#!/bin/bash
export PIRGPU_ALLOW_ENV=1
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4u; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config 4 --batch 16 --steps 6 --latency-runs 4 --no-cpu-baseline > $O/cfg4_profiled.json 2> /dev/null
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/stats_cfg4.csv 2>/dev/null
python3 $GRAFT_REPO_ROOT/tools/trace_summary.py $(find $O/prof -name "*kernel_trace.csv" | head -1) 20 > $O/trace_cfg4.txt
rm -rf $O/prof
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ntt_modes.py -m gpu -x -q -k looped 2>&1 | tail -3 > $O/tests.log
