#!/bin/bash
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4l; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/tests_all.log
timeout 300 python tools/soak.py 150 > $O/soak.log 2>&1
tail -3 $O/soak.log > $O/soak.tail
bash tools/r04_collect.sh
