#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
# round-3 experiment 3: GPU suite (pipelined rows step, multi-client), forced single-rank RCCL bench through the pipeline
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/exp3; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
sum() { python3 -c "
import json,sys
j=json.loads(open('$1').read().strip().splitlines()[-1])
print('$1', 'qps', round(j['value'],1), 'n_gpus', j['n_gpus'], 'rccl', j.get('rccl_ranks'), 'lat', j['latency_ms_single_query'], j.get('rows_step'), j.get('forced_dist_replies_equal_plain'), j.get('wire_process_request_ms'), j.get('multi_client_qps'))"; }
PIRGPU_FORCE_DIST=1 timeout 600 python3 bench.py --no-cpu-baseline --steps 50 > $O/forced.json 2> $O/forced.err; sum $O/forced.json; tail -3 $O/forced.err
PIRGPU_FORCE_DIST=1 PIRGPU_ROWS_PIPELINE=0 timeout 600 python3 bench.py --no-cpu-baseline --steps 50 > $O/forced_sync.json 2> $O/forced_sync.err; sum $O/forced_sync.json
timeout 600 python3 bench.py --no-cpu-baseline --steps 50 > $O/plain.json 2> $O/plain.err; sum $O/plain.json
timeout 300 python3 bench.py --gpus 2 > $O/gpus2.json 2> $O/gpus2.err; echo "gpus2 rc=$?"; tail -2 $O/gpus2.err
