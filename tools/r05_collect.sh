#!/bin/bash
# round 5, final collection on the GPU box (run through gpurun from the repo root): PMC traffic of the scan launch at the
# final scan source (bench.py refuses a stale file), the default bench line plain and under rocprofv3, the other
# configurations' reference lines, the whole GPU suite.  Results land in gpurun_out/final5/ and profiles/.
export PIRGPU_ALLOW_ENV=1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=r05
O=gpurun_out/final5
rm -rf $O; mkdir -p $O
COMMIT=${1:-unknown}
bash tools/pmc_scan_traffic.sh $O/${R}_pmc_scan_traffic.json $COMMIT 3 4 5 > $O/pmc_scan.log 2>&1
cp $O/${R}_pmc_scan_traffic.json profiles/${R}_pmc_scan_traffic.json
rm -rf gpurun_out/pmc_scan_cfg*_fetch gpurun_out/pmc_scan_cfg*_write
python3 bench.py > $O/${R}_bench.json 2> $O/bench.err
python3 bench.py --steps 20 --warmup 5 > $O/${R}_bench_driver_command.json 2> $O/bench_driver.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ${R} -- python3 bench.py --steps 20 --warmup 5 > $O/${R}_bench_profiled.json 2> $O/bench_prof.err
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/${R}_kernel_stats.csv
python3 tools/trace_summary.py $(find $O/prof -name "*kernel_trace.csv" | head -1) 1000 > $O/${R}_trace_summary.txt
rm -rf $O/prof
for c in 2 4 5; do python3 bench.py --config $c --batch 16 --steps 10 --no-cpu-baseline > $O/${R}_bench_cfg${c}_reference.json 2> $O/cfg$c.err; done
timeout 600 python tools/rank_budget.py --slots 3 1,2,4,8 > $O/budget_slots_cfg3.log 2>&1 && cp gpurun_out/rank_budget_slots_cfg3.json profiles/${R}_rank_budget_slots_cfg3.json
timeout 900 python tools/rank_budget.py --slots 4 2,4,8 > $O/budget_slots_cfg4.log 2>&1 && cp gpurun_out/rank_budget_slots_cfg4.json profiles/${R}_rank_budget_slots_cfg4.json
timeout 900 python tools/rank_budget.py --slots 5 4,8 > $O/budget_slots_cfg5.log 2>&1 && cp gpurun_out/rank_budget_slots_cfg5.json profiles/${R}_rank_budget_slots_cfg5.json
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_suite.log 2>&1
echo "suite rc=$?" > $O/rc.txt
tail -3 $O/gpu_suite.log
ls -la $O
