#!/bin/bash
# round 6, final collection on the GPU box (run through gpurun from the repo root, AFTER the kernels' last commit:
# `bash tools/r06_collect.sh <commit>`): PMC traffic of the scan launch and the per-kernel SQ / TCC counters at the final
# sources (both files are stamped with the sources' hashes; bench.py withholds a stale one), the default bench line plain,
# as the driver runs it, and under rocprofv3, the other configurations' reference lines and VALU tables, per-rank budgets of
# the slot-sharded step.  Results land in gpurun_out/final6/; what is to be judged is copied into profiles/.
export PIRGPU_ALLOW_ENV=1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=r06
O=gpurun_out/final6
mkdir -p $O
COMMIT=${1:-unknown}
PART=${2:-all}          # a: scan traffic, cfg 3 counters, bench lines; b: cfg 4 / 5 counters, per-rank budgets
export PIRGPU_PROFILED_COMMIT=$COMMIT
if [ $PART != b ]; then
bash tools/pmc_scan_traffic.sh $O/${R}_pmc_scan_traffic.json $COMMIT 3 4 5 > $O/pmc_scan.log 2>&1
cp $O/${R}_pmc_scan_traffic.json profiles/${R}_pmc_scan_traffic.json
rm -rf gpurun_out/pmc_scan_cfg*_fetch gpurun_out/pmc_scan_cfg*_write
bash tools/pmc_kernels.sh ${R}e > $O/pmc.log 2>&1
python3 tools/valu_roofline.py ${R}e $O/${R}_valu_roofline.json > $O/${R}_valu_table_cfg3.txt
cp gpurun_out/pmc_${R}e.json $O/${R}_pmc_counters.json
cp $O/${R}_valu_roofline.json profiles/${R}_valu_roofline.json
rm -rf gpurun_out/pmc_${R}e_valu gpurun_out/pmc_${R}e_lds gpurun_out/pmc_${R}e_fetch gpurun_out/pmc_${R}e_write
fi
if [ $PART != a ]; then
for c in 4 5; do
  bash tools/pmc_kernels.sh ${R}cfg$c --config $c --batch 8 --steps 1 --warmup 1 --latency-runs 2 --no-cpu-baseline > $O/pmc_cfg$c.log 2>&1
  python3 tools/valu_roofline.py ${R}cfg$c $O/${R}_valu_roofline_cfg$c.json > $O/${R}_valu_table_cfg$c.txt 2>&1
  rm -rf gpurun_out/pmc_${R}cfg${c}_valu gpurun_out/pmc_${R}cfg${c}_lds gpurun_out/pmc_${R}cfg${c}_fetch gpurun_out/pmc_${R}cfg${c}_write
done
fi
if [ $PART != b ]; then
python3 bench.py > $O/${R}_bench.json 2> $O/bench.err
python3 bench.py --steps 20 --warmup 5 > $O/${R}_bench_driver_command.json 2> $O/bench_driver.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ${R} -- python3 bench.py --steps 20 --warmup 5 > $O/${R}_bench_profiled.json 2> $O/bench_prof.err
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/${R}_kernel_stats.csv
python3 tools/trace_summary.py $(find $O/prof -name "*kernel_trace.csv" | head -1) 1000 > $O/${R}_trace_summary.txt
rm -rf $O/prof
for c in 2 4 5; do python3 bench.py --config $c --batch 16 --steps 10 --no-cpu-baseline > $O/${R}_bench_cfg${c}_reference.json 2> $O/cfg$c.err; done
fi
if [ $PART != a ]; then
timeout 600 python tools/rank_budget.py --slots 3 1,2,4,8 > $O/budget_slots_cfg3.log 2>&1 && cp gpurun_out/rank_budget_slots_cfg3.json $O/${R}_rank_budget_slots_cfg3.json
timeout 900 python tools/rank_budget.py --slots 4 8 > $O/budget_slots_cfg4.log 2>&1 && cp gpurun_out/rank_budget_slots_cfg4.json $O/${R}_rank_budget_slots_cfg4.json
timeout 900 python tools/rank_budget.py --slots 5 8 > $O/budget_slots_cfg5.log 2>&1 && cp gpurun_out/rank_budget_slots_cfg5.json $O/${R}_rank_budget_slots_cfg5.json
fi
ls -la $O
python3 - <<'PY'
import json
for n in ("r06_bench", "r06_bench_driver_command", "r06_bench_profiled", "r06_bench_cfg2_reference", "r06_bench_cfg4_reference", "r06_bench_cfg5_reference"):
    try:
        d = json.loads(open("gpurun_out/final6/%s.json" % n).read().strip().splitlines()[-1])
        print(n, round(d["value"], 1), d.get("latency_ms_single_query"), d["roofline"].get("frac"), d["roofline"].get("traffic_stale"),
              (d.get("roofline_compute") or {}).get("stale"), (d["roofline"].get("batch_launch") or {}).get("mean_ms"))
    except Exception as e:
        print(n, "ERR", e)
PY
