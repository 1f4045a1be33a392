#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
# usage: tools/ab.sh "ENV_A" "ENV_B" [rounds]   -- interleaved A/B of bench.py on the same box
A="$1"; B="$2"; R=${3:-3}
for i in $(seq $R); do
  for v in "$A" "$B"; do
    env $v python bench.py --no-cpu-baseline --batch 8 --steps 10 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$v', 'qps', round(j['value'],1), 'lat', j['latency_ms_single_query'], 'scan_ms', round(j['roofline']['kernel_ms'],4), 'frac', round(j['roofline']['frac'],3), j['phases_ms_single_query'])"
  done
done
