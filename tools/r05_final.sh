#!/bin/bash
# round 5, final verification (one gpurun call): the slots tests incl. the 24-group case, soaks in every arithmetic flavour
# (each mixes the slot-sharded step in), the 8-rank and 4-rank flows of bench.py on one GPU (all forms of the sharded step),
# the slot-sharded step through a real single-rank RCCL group
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5x; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_slots.py tests/test_gpu_distributed.py -q 2>&1 | tail -4 > $O/tests_slots.log
SOAK_SEED=12 timeout 500 python tools/soak.py 300 2>&1 | tail -1 > $O/soak_default.tail
SOAK_SEED=13 PIRGPU_NTT_MODE=0 timeout 400 python tools/soak.py 180 2>&1 | tail -1 > $O/soak_int.tail
SOAK_SEED=14 PIRGPU_NTT_MODE=2 timeout 400 python tools/soak.py 180 2>&1 | tail -1 > $O/soak_wide.tail
SOAK_SEED=15 PIRGPU_LOOP_TRANSFORMS=0 PIRGPU_FUSE_LAST=0 timeout 400 python tools/soak.py 120 2>&1 | tail -1 > $O/soak_unfused.tail
PIRGPU_BENCH_SHARE_GPU=1 timeout 900 python3 bench.py --gpus 8 --log-items 18 --steps 2 --warmup 1 --latency-runs 2 --no-cpu-baseline > $O/r05_bench_eight_ranks_sharing_one_gpu.json 2> $O/eight.err
PIRGPU_BENCH_SHARE_GPU=1 timeout 900 python3 bench.py --gpus 4 --log-items 18 --steps 2 --warmup 1 --latency-runs 2 --no-cpu-baseline > $O/r05_bench_four_ranks_sharing_one_gpu.json 2> $O/four.err
PIRGPU_FORCE_DIST=1 PIRGPU_EXCHANGE=slots timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r05_bench_forced_single_rank_rccl_slots.json 2> $O/forced.err
cat $O/*.tail $O/tests_slots.log
python3 - <<'PY'
import json
for f in ("eight", "four"):
    try:
        j = json.loads(open("gpurun_out/r5x/r05_bench_%s_ranks_sharing_one_gpu.json" % f).read().strip().splitlines()[-1])
        print(f, j["config"]["exchange"], j["exchange_autotune"]["ms_per_step"], j.get("slots_step", j["exchange_autotune"].get("slots_details", {})).get("row_sums_on_the_links"))
    except Exception as e:
        print(f, "ERR", e)
j = json.loads(open("gpurun_out/r5x/r05_bench_forced_single_rank_rccl_slots.json").read().strip().splitlines()[-1])
print("forced", j["value"], j.get("forced_dist_replies_equal_plain"))
PY
