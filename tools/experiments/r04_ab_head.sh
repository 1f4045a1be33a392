#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
# A/B of the head-stream expansion levels (PIRGPU_HEAD_LEVELS) on one box, interleaved
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi_client.py tests/test_gpu_distributed.py tests/test_gpu_mfma_scan.py -m gpu -x -q 2>&1 | tail -6 > $O/tests.log
for rep in 1 2; do
  for h in 0 5 3 4 6; do
    PIRGPU_BENCH_SKIP_WIRE=1 PIRGPU_HEAD_LEVELS=$h timeout 600 python bench.py --steps 100 --latency-runs 10 --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('head=$h rep=$rep value %.1f ms_per_step %.3f blocks %s' % (j['value'], j['ms_per_step'], j['timed_block_seconds']))" >> $O/ab.log
  done
done
cat $O/ab.log
