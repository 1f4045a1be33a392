#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/tests_all.log
PIRGPU_WIRE_TRACE=1 timeout 900 python bench.py --steps 100 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -c 3000 $O/bench.err > $O/bench.tail; rm -f $O/bench.err
PIRGPU_BENCH_WIRE_CALLERS=3 timeout 900 python bench.py --steps 60 --no-cpu-baseline > $O/bench_callers3.json 2> /dev/null
PIRGPU_BENCH_WIRE_CAPACITY=64 timeout 900 python bench.py --steps 60 --no-cpu-baseline > $O/bench_cap64.json 2> /dev/null
ls -la $O
