#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
# round 4, GPU call 1: new tests, default bench (wire legs), lane / worker experiments
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a; mkdir -p $O
python -m pytest tests/test_gpu_large_rings.py tests/test_gpu_multi_client.py tests/test_gpu_wire_extras.py tests/test_gpu_distributed.py -m gpu -x -q 2>&1 | tail -15 > $O/tests.log
PIRGPU_WIRE_TRACE=1 python bench.py --steps 100 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -c 6000 $O/bench.err > $O/bench.tail; rm -f $O/bench.err
PIRGPU_BENCH_WIRE_CAPACITY=64 python bench.py --steps 60 --no-cpu-baseline > $O/bench_cap64.json 2> /dev/null
PIRGPU_BENCH_WIRE_CALLERS=3 python bench.py --steps 60 --no-cpu-baseline > $O/bench_callers3.json 2> /dev/null
PIRGPU_LANES=3 python bench.py --steps 60 --workers 24 --no-cpu-baseline > $O/bench_lanes3.json 2> /dev/null
python bench.py --steps 20 --no-cpu-baseline > $O/bench_steps20.json 2> /dev/null
ls -la $O
