#!/bin/bash
# upper_fused at three waves per SIMD (plain variant, registers capped at 168, 96 workgroups per query) against the
# shipped LDS-twiddle variant at two waves per SIMD; library rebuilt per variant.  The capped build needs one line that
# is NOT in the tree (the experiment lost): `#ifdef PIRGPU_UPPER_WAVES` ->
# `__attribute__((amdgpu_waves_per_eu(PIRGPU_UPPER_WAVES)))` on upper_fused_kernel in ntt_kernels.hip.
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4y; mkdir -p $O
for rep in 1 2; do
  python -c "from pir_amd import build; build.build(force=True)" > $O/build_base.log 2>&1
  PIRGPU_BENCH_SKIP_WIRE=1 timeout 600 python3 bench.py --no-cpu-baseline > $O/cfg3_base_$rep.json 2> /dev/null
  PIRGPU_UPPER_LDS_TW=0 PIRGPU_BENCH_SKIP_WIRE=1 timeout 600 python3 bench.py --no-cpu-baseline > $O/cfg3_plain2_$rep.json 2> /dev/null
  PIRGPU_BUILD_DEFS="-DPIRGPU_UPPER_WAVES=3" python -c "from pir_amd import build; build.build(force=True)" > $O/build_w3.log 2>&1
  PIRGPU_UPPER_LDS_TW=0 PIRGPU_UPPER_BLOCKS_BATCH=96 PIRGPU_BENCH_SKIP_WIRE=1 timeout 600 python3 bench.py --no-cpu-baseline > $O/cfg3_plain3_b96_$rep.json 2> /dev/null
  PIRGPU_UPPER_LDS_TW=0 PIRGPU_BENCH_SKIP_WIRE=1 timeout 600 python3 bench.py --no-cpu-baseline > $O/cfg3_plain3_b64_$rep.json 2> /dev/null
done
python -c "from pir_amd import build; build.build(force=True)" > /dev/null 2>&1
python3 - <<'PY' > $O/summary.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/r4y/c*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f, round(d["value"],1), d.get("latency_ms_single_query"), d.get("phases_ms_single_query"))
PY
