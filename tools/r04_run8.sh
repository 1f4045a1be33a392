#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wire_extras.py tests/test_gpu_multi_client.py tests/test_gpu_client_roundtrip.py tests/test_cpp_facade.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 > $O/tests.log
for rep in 1 2; do
  timeout 900 python bench.py --steps 100 --no-cpu-baseline > $O/bench_$rep.json 2> /dev/null
done
( time PIRGPU_BENCH_SHARE_GPU=1 timeout 1500 python bench.py --gpus 8 --log-items 18 --steps 2 --warmup 1 --latency-runs 2 --no-cpu-baseline > $O/bench_share8.json 2> $O/bench_share8.err ) 2> $O/share8.time
tail -c 3000 $O/bench_share8.err > $O/bench_share8.tail; rm -f $O/bench_share8.err
( time PIRGPU_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 4 --log-items 18 --steps 2 --warmup 1 --latency-runs 2 --no-cpu-baseline > $O/bench_share4.json 2> /dev/null ) 2> $O/share4.time
