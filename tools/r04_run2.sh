#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
timeout 600 python tools/r04_debug_t42.py > $O/debug_t42.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ntt_modes.py -m gpu -x -q 2>&1 | tail -15 > $O/tests_parity.log
timeout 600 python bench.py --config 5 --batch 16 --steps 10 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 600 python bench.py --config 4 --batch 16 --steps 10 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
tail -c 1500 $O/bench_cfg5.err > $O/bench_cfg5.tail; rm -f $O/bench_cfg5.err $O/bench_cfg4.err
timeout 900 python -m pytest tests/test_gpu_full_size.py -m gpu -x -q -k "cfg5 or 5" 2>&1 | tail -8 > $O/tests_full5.log
ls -la $O
