#!/bin/bash
# round 6, GPU call 10: the default bench line with the library loaded before torch (system HIP runtime) and after it
# (torch's bundled runtime), alternating; smoke(); the new knob tests
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6j; mkdir -p $O
for rep in 1 2; do
  for tf in 0 1; do
    PIRGPU_BENCH_TORCH_FIRST=$tf python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_tf${tf}_$rep.json 2> $O/bench_tf${tf}_$rep.err
  done
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6j/bench_tf*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    w = d.get("wire_multi_client_qps", {})
    print(f.split("/")[-1], "qps", round(d["value"], 1), "lat", d["latency_ms_single_query"], "multi_client", d.get("multi_client_qps"),
          "wire", {k: w.get(k) for k in ("value", "single_caller", "single_caller_two_calls_in_flight")},
          "wire_single_ms", d.get("wire_process_request_ms"), d.get("hip_runtime"))
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 1500 python -m pytest tests/test_gpu_ntt_modes.py -x -q -m gpu -k round6 > $O/tests_knobs.log 2>&1; tail -3 $O/tests_knobs.log
