#!/bin/bash
# final verification + collection of the round (one gpurun call): GPU suite, soaks in every arithmetic flavour, the
# 8-rank flow on one GPU, then tools/experiments/r04_collect.sh
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4x; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/tests_all.log
SOAK_SEED=2 timeout 300 python tools/soak.py 120 2>&1 | tail -1 > $O/soak_default.tail
SOAK_SEED=3 PIRGPU_NTT_MODE=0 timeout 300 python tools/soak.py 100 2>&1 | tail -1 > $O/soak_int.tail
SOAK_SEED=4 PIRGPU_NTT_MODE=2 timeout 300 python tools/soak.py 100 2>&1 | tail -1 > $O/soak_wide.tail
SOAK_SEED=5 PIRGPU_LOOP_TRANSFORMS=0 timeout 300 python tools/soak.py 80 2>&1 | tail -1 > $O/soak_noloop.tail
PIRGPU_BENCH_SHARE_GPU=1 timeout 900 python3 bench.py --gpus 8 --log-items 18 --steps 2 --warmup 1 --latency-runs 2 --no-cpu-baseline > $O/r04_bench_eight_ranks_sharing_one_gpu.json 2> $O/eight.err
sed -i "s/r04_pmc_scan_traffic.json [0-9a-f]\{7\} 3 4 5/r04_pmc_scan_traffic.json b192b5e 3 4 5/" tools/experiments/r04_collect.sh
bash tools/experiments/r04_collect.sh
cp $O/r04_bench_eight_ranks_sharing_one_gpu.json gpurun_out/final/
