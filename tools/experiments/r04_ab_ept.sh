#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
# A/B of the N = 16384 transform organisation on one box: 32 residues per thread in 512-thread workgroups (with / without
# twiddle prefetch) against 16 residues per thread in 1024-thread workgroups; rebuilds the library for each variant.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
# (the shipped default is 16 residues per thread, PIRGPU_LOG_EPT14 = 4 in device_params.h: the two 32-residue arms have to
# ask for theirs, and at 16 residues per thread the prefetch switch has no effect)
for v in "ept32_pf1:-DPIRGPU_LOG_EPT14=5" "ept16:" "ept32_pf0:-DPIRGPU_LOG_EPT14=5 -DPIRGPU_PF14=0" ; do
  tag=${v%%:*}; defs=${v#*:}
  PIRGPU_BUILD_DEFS="$defs" python -c "from pir_amd import build; build.build(force=True)" > $O/build_$tag.log 2>&1
  for rep in 1 2; do
    timeout 600 python bench.py --config 5 --batch 16 --steps 10 --no-cpu-baseline > $O/cfg5_${tag}_$rep.json 2> /dev/null
  done
  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$tag -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --batch 16 --steps 4 --latency-runs 6 --no-cpu-baseline > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
  cp $(find $O/prof_$tag -name "*kernel_stats.csv" | head -1) $O/stats_$tag.csv 2>/dev/null
  rm -rf $O/prof_$tag
done
python -c "from pir_amd import build; build.build(force=True)" > /dev/null 2>&1
ls $O
