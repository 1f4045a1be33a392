#!/bin/bash
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4hl; mkdir -p $O
for rep in 1 2 3; do
  for p in 0 -1; do
    PIRGPU_HEAD_PRIORITY=$p python3 tools/experiments/r04_wire_load.py 2 30 2>&1 | grep callers > $O/load_p${p}_$rep.txt
  done
done
for p in 0 -1; do PIRGPU_HEAD_PRIORITY=$p python3 tools/experiments/r04_wire_load.py 1 30 2>&1 | grep callers > $O/load1_p${p}.txt; done
cat $O/load_p0_*.txt $O/load_p-1_*.txt $O/load1_*.txt > $O/summary.txt
