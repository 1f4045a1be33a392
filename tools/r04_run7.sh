#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wire_extras.py tests/test_gpu_multi_client.py tests/test_gpu_client_roundtrip.py tests/test_gpu_large_rings.py -m gpu -x -q 2>&1 | tail -5 > $O/tests.log
for rep in 1 2; do
  timeout 900 python bench.py --steps 100 --no-cpu-baseline > $O/bench_$rep.json 2> /dev/null
done
python tools/r04_wire_load.py 2 20 2>&1 | grep callers > $O/wire_load.log
python tools/r04_wire_load.py 4 20 2>&1 | grep callers >> $O/wire_load.log
