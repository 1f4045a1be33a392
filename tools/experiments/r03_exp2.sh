#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
# round-3 experiment 2: full GPU suite (key sets, wide scan), bench with multi-client extra, cfg 3 wide vs narrow
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/exp2; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
sum() { python3 -c "
import json,sys
j=json.loads(open('$1').read().strip().splitlines()[-1])
print('$1', 'qps', round(j['value'],1), 'lat', j['latency_ms_single_query'], 'scan_ms', round(j['roofline']['kernel_ms'],4), 'frac', round(j['roofline']['frac'],3), j['roofline']['kernel'][:40], j['phases_ms_single_query'], j.get('wire_process_request_ms'), j.get('multi_client_qps'))"; }
python3 bench.py --no-cpu-baseline --steps 50 > $O/cfg3.json 2> $O/cfg3.err; sum $O/cfg3.json
PIRGPU_SCAN_MFMA_WIDE=1 python3 bench.py --no-cpu-baseline --steps 50 > $O/cfg3_wide.json 2> $O/cfg3_wide.err; sum $O/cfg3_wide.json
python3 bench.py --no-cpu-baseline --steps 50 > $O/cfg3_b.json 2> $O/cfg3_b.err; sum $O/cfg3_b.json
