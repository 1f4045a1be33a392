#!/bin/bash
# round 6, GPU call 11: the default bench line on the system HIP runtime (no torch in the process) against torch's bundled
# runtime, alternating
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6k; mkdir -p $O
for rep in 1 2 3; do
  for tf in 0 1; do
    PIRGPU_BENCH_TORCH_FIRST=$tf python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_tf${tf}_$rep.json 2> $O/bench_tf${tf}_$rep.err
  done
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6k/bench_tf*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:]); continue
    w = d.get("wire_multi_client_qps", {})
    wr = d.get("wire_process_request_ms", {})
    print(f.split("/")[-1], "qps", round(d["value"], 1), "lat", d["latency_ms_single_query"], "multi_client", round(d["multi_client_qps"]["value"], 1),
          "wire2", round(w.get("value", 0), 1), "single", round(w.get("single_caller", {}).get("value", 0), 1),
          "begin_end", round(w.get("single_caller_two_calls_in_flight", {}).get("value", 0), 1),
          "lone_ms", wr.get("repeat_at_c_abi"), "new_client_ms", wr.get("new_client_seeded_keys_ms"), d.get("hip_runtime"))
PY
