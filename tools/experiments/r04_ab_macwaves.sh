#!/bin/bash
# product kernels (ks_mac_intt / ks_mac_combine / ks_last_ntt) with more registers and fewer waves: -DPIRGPU_MAC_MAXWAVES=3
# (132-142 VGPRs) and =2 (169) against the default (126-128, four waves per SIMD); library rebuilt per variant.
# The macro is NOT in the tree (the experiment lost): `__attribute__((amdgpu_waves_per_eu(1,
# PIRGPU_MAC_MAXWAVES)))` on the three kernels in ntt_kernels.hip.
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4z; mkdir -p $O
for rep in 1 2 3; do
for v in "w4:" "w3:-DPIRGPU_MAC_MAXWAVES=3" "w2:-DPIRGPU_MAC_MAXWAVES=2"; do
  tag=${v%%:*}; defs=${v#*:}
  PIRGPU_BUILD_DEFS="$defs" python -c "from pir_amd import build; build.build(force=True)" > $O/build_$tag.log 2>&1
  PIRGPU_BENCH_SKIP_WIRE=1 timeout 600 python3 bench.py --no-cpu-baseline > $O/cfg3_${tag}_$rep.json 2> /dev/null
done
done
python -c "from pir_amd import build; build.build(force=True)" > /dev/null 2>&1
python3 - <<'PY' > $O/summary.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/r4z/c*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f, round(d["value"],1), d.get("latency_ms_single_query"), d.get("phases_ms_single_query"))
PY
