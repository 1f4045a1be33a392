#!/bin/bash
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4j; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/tests_all.log
timeout 900 python bench.py --steps 100 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.err > $O/bench.tail; rm -f $O/bench.err
timeout 600 python bench.py --steps 20 --no-cpu-baseline > $O/bench_steps20.json 2> /dev/null
