#!/bin/bash
# A/B of the looped transform kernels' register budget on one box: 128 VGPRs (four waves per SIMD, no twiddle prefetch
# inside the loop) against the first form (prefetch on, 131-142 VGPRs: three waves per SIMD); library rebuilt per variant.
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4v; mkdir -p $O
for rep in 1 2 3; do
for v in "cap128:" "pf138:-DPIRGPU_PF_LOOP=1"; do
  tag=${v%%:*}; defs=${v#*:}
  PIRGPU_BUILD_DEFS="$defs" python -c "from pir_amd import build; build.build(force=True)" > $O/build_$tag.log 2>&1
  PIRGPU_BENCH_SKIP_WIRE=1 timeout 600 python3 bench.py --no-cpu-baseline > $O/cfg3_${tag}_$rep.json 2> /dev/null
  timeout 600 python3 bench.py --config 4 --batch 16 --steps 10 --no-cpu-baseline > $O/cfg4_${tag}_$rep.json 2> /dev/null
done
done
python -c "from pir_amd import build; build.build(force=True)" > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_ntt_modes.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 > $O/tests.log
python3 - <<'PY' > $O/summary.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/r4v/c*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f, round(d["value"],1), d.get("latency_ms_single_query"))
PY
