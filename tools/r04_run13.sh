#!/bin/bash
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4n; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ntt_modes.py tests/test_gpu_full_size.py tests/test_gpu_large_rings.py -m gpu -x -q 2>&1 | tail -8 > $O/tests_loop.log
bash tools/r04_ab_loop.sh
bash tools/r04_run12.sh
