#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_large_rings.py tests/test_gpu_multi_client.py tests/test_gpu_wire_extras.py tests/test_gpu_client_roundtrip.py -m gpu -x -q 2>&1 | tail -12 > $O/tests.log
PIRGPU_WIRE_TRACE=1 timeout 900 python bench.py --steps 100 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -c 5000 $O/bench.err > $O/bench.tail; rm -f $O/bench.err
bash tools/r04_ab_ept.sh
