#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
# A/B of the scan's fp64 fold and of paired (16-query) database passes on one box, interleaved
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4k; mkdir -p $O
PIRGPU_SCAN_F64_FOLD=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mfma_scan.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -5 > $O/tests_fold.log
PIRGPU_SCAN_F64_FOLD=1 PIRGPU_SCAN_PAIR=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi_client.py tests/test_gpu_mfma_scan.py tests/test_gpu_wire_extras.py -m gpu -x -q 2>&1 | tail -5 > $O/tests_fold_pair.log
for rep in 1 2; do
  for v in "0 0 16" "1 0 16" "1 1 32" "0 1 32"; do
    set -- $v
    PIRGPU_BENCH_SKIP_WIRE=1 PIRGPU_SCAN_F64_FOLD=$1 PIRGPU_SCAN_PAIR=$2 timeout 600 python bench.py --steps 100 --workers $3 --latency-runs 10 --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fold=$1 pair=$2 rep=$rep value %.1f ms_per_step %.3f single_scan_ms %.4f batch_launch %s' % (j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], {k: v for k, v in (j['roofline'].get('batch_launch') or {}).items() if k in ('mean_ms','min_ms','launches','workgroups')}))" >> $O/ab.log
  done
done
for c in 4 5; do
  for f in 0 1; do
    PIRGPU_SCAN_F64_FOLD=$f timeout 600 python bench.py --config $c --batch 16 --steps 10 --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg$c fold=$f value %.1f lat %.3f scan_ms %.4f' % (j['value'], j['latency_ms_single_query'], j['roofline']['kernel_ms']))" >> $O/ab.log
  done
done
cat $O/ab.log
