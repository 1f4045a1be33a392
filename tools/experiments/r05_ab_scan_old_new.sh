#!/bin/bash
# same-box A/B of the scan kernel before (round 4: .ab_old/, built from commit a60ca13) and after the groups / slot-range
# generalisation: single-query scan launch and batched throughput at cfg 3, 4, 5, alternating
export PIRGPU_ALLOW_ENV=1 PIRGPU_BENCH_SKIP_WIRE=1 PIRGPU_BENCH_SKIP_SWEEP=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_scan; mkdir -p $O
for rep in 1 2; do
  for cfg in 3 4 5; do
    for v in old new; do
      if [ $v = old ]; then d=.ab_old; else d=.; fi
      (cd $d && timeout 600 python bench.py --config $cfg --batch 16 --steps 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/cfg${cfg}_${v}_$rep.json 2> /dev/null)
    done
  done
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/ab_scan/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value'], 1), 'scan_ms', round(d['roofline']['kernel_ms'], 4), 'lat', d.get('latency_ms_single_query'))
    except Exception as e:
        print(f, 'ERR', e)
PY
