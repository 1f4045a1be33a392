#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
# round-3 experiment 1: full GPU suite, then wide (4-wave) vs 8-wave MFMA scan on cfg 3/4/5
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/exp1; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
sum() { python3 -c "
import json,sys
j=json.loads(open('$1').read().strip().splitlines()[-1])
print('$1', 'qps', round(j['value'],1), 'lat', j['latency_ms_single_query'], 'scan_ms', round(j['roofline']['kernel_ms'],4), 'frac', round(j['roofline']['frac'],3), j['roofline']['kernel'][:40], j['phases_ms_single_query'])"; }
python3 bench.py --no-cpu-baseline --steps 50 > $O/cfg3.json 2> $O/cfg3.err; sum $O/cfg3.json
for c in 4 5; do
  for w in auto 0; do
    if [ $w = auto ]; then unset PIRGPU_SCAN_MFMA_WIDE; else export PIRGPU_SCAN_MFMA_WIDE=$w; fi
    timeout 900 python3 bench.py --config $c --batch 16 --steps 10 --no-cpu-baseline > $O/cfg${c}_wide$w.json 2> $O/cfg${c}_wide$w.err; sum $O/cfg${c}_wide$w.json
  done
done
