#!/bin/bash
# SQ / LDS counters of the N = 16384 kernels (cfg 5): what bounds a 16384-point transform with one workgroup per CU
export PIRGPU_ALLOW_ENV=1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/pmc_kernels.sh r04cfg5 --config 5 --batch 8 --steps 1 --warmup 1 --latency-runs 2 --no-cpu-baseline > gpurun_out/pmc_r04cfg5.log 2>&1
python3 tools/valu_roofline.py r04cfg5 gpurun_out/r04_valu_roofline_cfg5.json > gpurun_out/r04_valu_table_cfg5.txt 2>&1
rm -rf gpurun_out/pmc_r04cfg5_valu gpurun_out/pmc_r04cfg5_lds gpurun_out/pmc_r04cfg5_fetch gpurun_out/pmc_r04cfg5_write
