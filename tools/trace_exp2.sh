cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for V in 0 1 3 4 8 16 31; do
export PIRGPU_EXP_VARIANT=$V
T=exp_v$V
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$T -- python3 bench.py --no-cpu-baseline --steps 6 --latency-runs 2 > gpurun_out/$T.json 2>gpurun_out/$T.err
python tools/trace_summary.py $(ls gpurun_out/$T/*/*kernel_trace.csv | head -1) 500 > gpurun_out/${T}_summary.txt
rm -rf gpurun_out/$T
echo "variant $V: $(grep 'ks_last_ntt_kernel<1, true>  *8192' gpurun_out/${T}_summary.txt)"
done
