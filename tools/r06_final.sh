#!/bin/bash
# round 6, final verification (one gpurun call): the whole GPU suite, smoke(), soaks in every arithmetic flavour, the 8-rank
# and 2-rank flows of bench.py on one GPU (candidate order, watchdog slices), the slot-sharded step through a real single-rank
# RCCL group
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6x; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_suite.log 2>&1
echo "suite rc=$?" > $O/rc.txt; tail -3 $O/gpu_suite.log
SOAK_SEED=21 timeout 500 python tools/soak.py 240 2>&1 | tail -1 > $O/soak_default.tail
SOAK_SEED=22 PIRGPU_NTT_MODE=0 timeout 400 python tools/soak.py 120 2>&1 | tail -1 > $O/soak_int.tail
SOAK_SEED=23 PIRGPU_NTT_MODE=2 timeout 400 python tools/soak.py 120 2>&1 | tail -1 > $O/soak_wide.tail
SOAK_SEED=24 PIRGPU_C0_NTT=2 timeout 400 python tools/soak.py 120 2>&1 | tail -1 > $O/soak_c0ntt.tail
SOAK_SEED=25 PIRGPU_PACK_BYTES=7 timeout 400 python tools/soak.py 120 2>&1 | tail -1 > $O/soak_pack7.tail
PIRGPU_BENCH_SHARE_GPU=1 timeout 900 python3 bench.py --gpus 8 --log-items 18 --steps 2 --warmup 1 --latency-runs 2 --no-cpu-baseline > $O/r06_bench_eight_ranks_sharing_one_gpu.json 2> $O/eight.err
PIRGPU_FORCE_DIST=1 PIRGPU_EXCHANGE=slots timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r06_bench_forced_single_rank_rccl_slots.json 2> $O/forced.err
cat $O/*.tail
python3 - <<'PY'
import json
try:
    j = json.loads(open("gpurun_out/r6x/r06_bench_eight_ranks_sharing_one_gpu.json").read().strip().splitlines()[-1])
    t = j["exchange_autotune"]
    print("eight", j["config"]["exchange"], t["order"], t["ms_per_step"], t["wall_s"], t["watchdog_slice_s"])
except Exception as e:
    print("eight ERR", e)
try:
    j = json.loads(open("gpurun_out/r6x/r06_bench_forced_single_rank_rccl_slots.json").read().strip().splitlines()[-1])
    print("forced", j["value"], j.get("forced_dist_replies_equal_plain"), j.get("hip_runtime"))
except Exception as e:
    print("forced ERR", e)
PY
