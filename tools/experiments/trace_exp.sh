export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=$1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$T -- python3 bench.py --no-cpu-baseline --steps 20 > gpurun_out/$T.json 2>gpurun_out/$T.err
python tools/trace_summary.py $(ls gpurun_out/$T/*/*kernel_trace.csv | head -1) 2000 > gpurun_out/${T}_summary.txt
rm -rf gpurun_out/$T
grep "ks_last_ntt\|ks_mac_combine\|upper_fused\|ks_digit_kernel<1, true>  *12288" gpurun_out/${T}_summary.txt | head -8
python -c "
import json;d=json.loads(open('gpurun_out/$T.json').read().strip().splitlines()[-1]);print('qps',d['value'])"
