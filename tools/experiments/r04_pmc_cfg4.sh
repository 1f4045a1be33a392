#!/bin/bash
# SQ / LDS counters of the N = 8192 kernels (cfg 4)
export PIRGPU_ALLOW_ENV=1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/pmc_kernels.sh r04cfg4 --config 4 --batch 8 --steps 1 --warmup 1 --latency-runs 2 --no-cpu-baseline > gpurun_out/pmc_r04cfg4.log 2>&1
python3 tools/valu_roofline.py r04cfg4 gpurun_out/r04_valu_roofline_cfg4.json > gpurun_out/r04_valu_table_cfg4.txt 2>&1
rm -rf gpurun_out/pmc_r04cfg4_valu gpurun_out/pmc_r04cfg4_lds gpurun_out/pmc_r04cfg4_fetch gpurun_out/pmc_r04cfg4_write
