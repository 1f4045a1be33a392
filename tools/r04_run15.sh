#!/bin/bash
export PIRGPU_ALLOW_ENV=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4p; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_ntt_modes.py tests/test_gpu_parity.py tests/test_gpu_large_rings.py tests/test_gpu_full_size.py tests/test_gpu_mfma_scan.py -m gpu -x -q 2>&1 | tail -8 > $O/tests_exchange.log
bash tools/r04_ab_exchange.sh
