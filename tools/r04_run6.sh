#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4f; mkdir -p $O
for rep in 1 2; do
  for h in 0 5; do
    for c in 2 1 3; do
      echo "head=$h callers=$c rep=$rep" >> $O/wire_ab.log
      PIRGPU_HEAD_LEVELS=$h python tools/r04_wire_load.py $c 20 2>&1 | grep callers >> $O/wire_ab.log
    done
  done
done
PIRGPU_HEAD_LEVELS=0 python bench.py --steps 60 --no-cpu-baseline > $O/bench_head0.json 2> /dev/null
PIRGPU_HEAD_LEVELS=5 python bench.py --steps 60 --no-cpu-baseline > $O/bench_head5.json 2> /dev/null
cat $O/wire_ab.log
