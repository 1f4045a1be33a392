#!/bin/bash
# round 6, GPU call 9: 6 / 7-byte intermediates again (product kernels held at 128 registers at N = 8192; tree in doubles at
# N = 16384), against doubles; the large-ring + full-size tests with the packed forms
cd ${GRAFT_REPO_ROOT:-.}
export PIRGPU_ALLOW_ENV=1
O=gpurun_out/r6i; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_large_rings.py tests/test_gpu_ntt_modes.py -x -q -m gpu > $O/tests.log 2>&1
tail -2 $O/tests.log
tools/experiments/r06_ab.sh $O 3 4 "--batch 16 --steps 5 --warmup 2" head:PIRGPU_PACK_BYTES=8 head > $O/summary_cfg4.txt 2>&1
cut -c1-200 $O/summary_cfg4.txt
tools/experiments/r06_ab.sh $O 2 5 "--batch 16 --steps 3 --warmup 1" head:PIRGPU_PACK_BYTES=8 head head:PIRGPU_TREE40_WIDE=1 > $O/summary_cfg5.txt 2>&1
cut -c1-200 $O/summary_cfg5.txt
timeout 1500 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu > $O/tests_full.log 2>&1
tail -2 $O/tests_full.log
bash tools/experiments/r06_run6.sh _torch
PIRGPU_TRACE_NO_TORCH=1 bash tools/experiments/r06_run6.sh _notorch
