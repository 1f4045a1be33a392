#!/bin/bash
# A/B of the wave-local LDS exchanges (ntt_core.h, exchange_sync) on one box: library rebuilt per variant.
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4p; mkdir -p $O
for rep in 1 2; do
for v in "local:" "barrier:-DPIRGPU_WAVE_LOCAL_EXCHANGE=0"; do
  tag=${v%%:*}; defs=${v#*:}
  PIRGPU_BUILD_DEFS="$defs" python -c "from pir_amd import build; build.build(force=True)" > $O/build_$tag.log 2>&1
  PIRGPU_BENCH_SKIP_WIRE=1 timeout 600 python3 bench.py --no-cpu-baseline > $O/cfg3_${tag}_$rep.json 2> /dev/null
  timeout 600 python3 bench.py --config 4 --batch 16 --steps 10 --no-cpu-baseline > $O/cfg4_${tag}_$rep.json 2> /dev/null
  timeout 600 python3 bench.py --config 5 --batch 16 --steps 10 --no-cpu-baseline > $O/cfg5_${tag}_$rep.json 2> /dev/null
  timeout 600 python3 bench.py --config 2 --batch 16 --steps 10 --no-cpu-baseline > $O/cfg2_${tag}_$rep.json 2> /dev/null
done
done
python -c "from pir_amd import build; build.build(force=True)" > /dev/null 2>&1
python3 - <<'PY' > $O/summary.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/r4p/c*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f, round(d["value"],1), d.get("latency_ms_single_query"), d.get("phases_ms_single_query"))
PY
