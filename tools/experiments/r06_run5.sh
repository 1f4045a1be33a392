#!/bin/bash
# round 6, GPU call 5: where the c0-in-NTT-form tree loses (per-kernel trace of the step in both forms), the
# short-transform prototype, the whole GPU suite at the current tree
cd ${GRAFT_REPO_ROOT:-.}
export PIRGPU_ALLOW_ENV=1 PIRGPU_BENCH_SKIP_WIRE=1 PIRGPU_BENCH_SKIP_SWEEP=1
O=gpurun_out/r6e; mkdir -p $O
for e in 4 3 2; do tools/ntt_short_proto_$e > $O/short_proto_ept$e.txt 2>&1; done
cat $O/short_proto_ept*.txt | grep -E "residues|prefetch 1"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for m in 0 2 1; do
  PIRGPU_C0_NTT=$m rocprofv3 --kernel-trace --output-format csv -d $O/prof$m -o t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c0ntt$m.json 2> $O/bench_c0ntt$m.err
  python3 tools/trace_summary.py $(find $O/prof$m -name "*kernel_trace.csv" | head -1) 20000 > $O/trace_c0ntt$m.txt
  rm -rf $O/prof$m
done
head -40 $O/trace_c0ntt0.txt; head -44 $O/trace_c0ntt2.txt
timeout 1800 python -m pytest tests -m gpu -q -x > $O/gpu_suite.log 2>&1
tail -4 $O/gpu_suite.log
