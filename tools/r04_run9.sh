#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4i; mkdir -p $O
PIRGPU_SCAN_PAIR=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi_client.py tests/test_gpu_mfma_scan.py tests/test_gpu_wire_extras.py -m gpu -x -q 2>&1 | tail -8 > $O/tests_pair.log
python tools/soak.py 40 > $O/soak_pair0.log 2>&1
PIRGPU_SCAN_PAIR=1 python tools/soak.py 120 > $O/soak_pair.log 2>&1
for rep in 1 2; do
  for pr in 0 1; do
    PIRGPU_BENCH_SKIP_WIRE=1 PIRGPU_SCAN_PAIR=$pr timeout 600 python bench.py --steps 100 --latency-runs 10 --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pair=$pr rep=$rep value %.1f ms_per_step %.3f workers %s' % (j['value'], j['ms_per_step'], j['config']['workers']))" >> $O/ab.log
  done
done
PIRGPU_BENCH_SKIP_WIRE=1 PIRGPU_SCAN_PAIR=1 PIRGPU_SCAN_MFMA_WGS_BATCH=96 timeout 600 python bench.py --steps 100 --latency-runs 10 --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pair=1 wgs=96 value %.1f' % j['value'])" >> $O/ab.log
PIRGPU_BENCH_SKIP_WIRE=1 PIRGPU_SCAN_PAIR=1 PIRGPU_SCAN_MFMA_WGS_BATCH=160 timeout 600 python bench.py --steps 100 --latency-runs 10 --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pair=1 wgs=160 value %.1f' % j['value'])" >> $O/ab.log
cat $O/ab.log
