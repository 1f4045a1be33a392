#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/exp4; rm -rf $O; mkdir -p $O
f() { grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; }
echo "== rccl check full size (chunked all_to_all)"; python3 tools/experiments/r03_rccl_check.py 20 64 release 2> $O/c1.err | f
echo "== rccl check, every collective forced piecewise (1 MB)"; PIRGPU_MAX_COLLECTIVE_MB=1 python3 tools/experiments/r03_rccl_check.py 16 24 release 2> $O/c2.err | f
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
sum() { python3 -c "
import json,sys
j=json.loads(open('$1').read().strip().splitlines()[-1])
print('$1', 'qps', round(j['value'],1), 'n_gpus', j['n_gpus'], 'rccl', j.get('rccl_ranks'), 'lat', j['latency_ms_single_query'], j.get('rows_step'), 'forced_equal', j.get('forced_dist_replies_equal_plain'))"; }
PIRGPU_FORCE_DIST=1 timeout 600 python3 bench.py --no-cpu-baseline --steps 50 > $O/forced.json 2> $O/forced.err; sum $O/forced.json
PIRGPU_FORCE_DIST=1 PIRGPU_ROWS_PIPELINE=0 timeout 600 python3 bench.py --no-cpu-baseline --steps 50 > $O/forced_sync.json 2> $O/forced_sync.err; sum $O/forced_sync.json
