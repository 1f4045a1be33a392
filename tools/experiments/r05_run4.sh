#!/bin/bash
# round 5, fourth GPU run: the slot-sharded step at the real sizes of cfg 4 / cfg 5 (full-size tests) and the default bench
# line (reference sweep and C-ABI request timing inside)
export PIRGPU_ALLOW_ENV=1
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_slots.py -q > gpurun_out/r05_t_full.log 2>&1
echo "full rc=$?" > gpurun_out/r05_run4_rc.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench1.json 2> gpurun_out/r05_bench1.err
echo "bench rc=$?" >> gpurun_out/r05_run4_rc.txt
cat gpurun_out/r05_run4_rc.txt; tail -5 gpurun_out/r05_t_full.log
