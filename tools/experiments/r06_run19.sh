#!/bin/bash
# round 6, GPU call 19: every form of the sharded step through a REAL single-rank RCCL process group (the collective code
# paths with one rank): replies must equal the plain ones
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6s; mkdir -p $O
for ex in packed replicated slots u64; do
  PIRGPU_FORCE_DIST=1 timeout 600 python bench.py --exchange $ex --steps 10 --warmup 2 --no-cpu-baseline > $O/forced_$ex.json 2> $O/forced_$ex.err
done
PIRGPU_FORCE_DIST=both timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/forced_both.json 2> $O/forced_both.err
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6s/forced_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(j["value"], 1), j["config"].get("exchange"), j.get("forced_dist_replies_equal_plain"), j.get("rccl_ranks"), "replicas" in j.get("replicas_reference", {}) or j.get("replicas_reference", {}).get("value"))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-500:])
PY
