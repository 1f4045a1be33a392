#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
# Per-kernel SQ / TCC counters of the whole query path from rocprofv3 PMC passes (kernel trace only, one
# counter group per pass; FETCH_SIZE and WRITE_SIZE cannot share a pass).  Run on the GPU box from the repo root:
#   bash tools/pmc_kernels.sh [tag] [bench args...]  -> gpurun_out/pmc_<tag>_<group>/, gpurun_out/pmc_<tag>.json
# The JSON holds, per kernel name, the mean of every counter over its dispatches plus the dispatch count;
# tools/pmc_summarise.py turns it into the VALU-issue roofline table of DESIGN.md section 4.
export TMPDIR=/tmp
TAG=${1:-r02}
shift
ARGS=${@:---no-cpu-baseline --steps 3 --warmup 1 --latency-runs 4}
declare -A CGRP
CGRP[valu]="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
CGRP[lds]="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"
CGRP[fetch]="FETCH_SIZE"
CGRP[write]="WRITE_SIZE"
for g in valu lds fetch write; do
  d=gpurun_out/pmc_${TAG}_$g
  rm -rf $d
  rocprofv3 --pmc ${CGRP[$g]} --kernel-trace --output-format csv -d $d -o pmc -- \
    python3 bench.py $ARGS > $d.log 2>&1
done
python3 tools/pmc_summarise.py $TAG
