#!/bin/bash
# fuse_mac_nodes (tree ciphertexts per launch from which ks_mac_combine replaces ks_mac_intt + ks_combine): 128 (default)
# against 64 / 32 / 16, single-query latency and batch throughput, alternating
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4fm; mkdir -p $O
for rep in 1 2 3; do
  for f in 128 64 32 16; do
    PIRGPU_FUSE_MAC_NODES=$f PIRGPU_BENCH_SKIP_WIRE=1 timeout 600 python3 bench.py --no-cpu-baseline --latency-runs 100 > $O/cfg3_fm${f}_$rep.json 2> /dev/null
  done
done
for f in 128 32; do
  PIRGPU_FUSE_MAC_NODES=$f timeout 600 python3 bench.py --config 2 --batch 16 --steps 10 --no-cpu-baseline --latency-runs 40 > $O/cfg2_fm${f}.json 2> /dev/null
  PIRGPU_FUSE_MAC_NODES=$f timeout 600 python3 bench.py --config 4 --batch 16 --steps 10 --no-cpu-baseline > $O/cfg4_fm${f}.json 2> /dev/null
done
python3 - <<'PY' > $O/summary.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/r4fm/c*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f, round(d["value"],1), d.get("latency_ms_single_query"), d.get("phases_ms_single_query"))
PY
