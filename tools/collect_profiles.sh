#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
# Collects the round's committed profile artifacts on the GPU box (run through gpurun from the repo root):
# PMC passes -> VALU roofline (bench.py embeds it), plain + rocprofv3-profiled default bench, cfg 2/4/5 reference runs.
# Results land in gpurun_out/final/; copy what is to be judged into profiles/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${1:-r03}            # round tag of the file names
O=gpurun_out/final
rm -rf $O; mkdir -p $O
bash tools/pmc_kernels.sh ${R}e > $O/pmc.log 2>&1
python3 tools/valu_roofline.py ${R}e $O/${R}_valu_roofline.json > $O/valu_table.txt
cp gpurun_out/pmc_${R}e.json $O/${R}_pmc_counters.json
cp $O/${R}_valu_roofline.json profiles/${R}_valu_roofline.json
rm -rf gpurun_out/pmc_${R}e_valu gpurun_out/pmc_${R}e_lds gpurun_out/pmc_${R}e_fetch gpurun_out/pmc_${R}e_write
python3 bench.py > $O/${R}_bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ${R} -- python3 bench.py > $O/${R}_bench_profiled.json 2> $O/bench_prof.err
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/${R}_kernel_stats.csv
python3 tools/trace_summary.py $(find $O/prof -name "*kernel_trace.csv" | head -1) 1000 > $O/${R}_trace_summary.txt
rm -rf $O/prof
for c in 2 4 5; do python3 bench.py --config $c --batch 16 --steps 10 --no-cpu-baseline > $O/${R}_bench_cfg${c}_reference.json 2> $O/cfg$c.err; done
ls -la $O
