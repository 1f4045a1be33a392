"""Which engine carries the wire path's reply downloads?  64 clients' requests through pirgpu_process_requests, `calls`
times, meant to run under `rocprofv3 --kernel-trace --memory-copy-trace`; prints the wall-clock window of the timed
calls (ns, the tracer's clock) so that the trace can be cut to it (tools/experiments/r06_run6.sh)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if os.environ.get("PIRGPU_TRACE_NO_TORCH") != "1":
    import torch  # noqa  (what bench.py has loaded when it runs the wire legs)
import pir_amd, bench
import seal_wire as W
class A: pass
args = A(); args.config = 3; args.log_items = 20; args.dims = 2
enc, pp, _ = bench.build_workload(args, pir_amd)
NCL = 64
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 10
raw, keys, queries = bench.synthetic_inputs(pp, n_queries=NCL)
db = pir_amd.PIRDatabase.Create(pp); db.populate(raw); db.finalize(release_staging=True)
srv = pir_amd.PIRServer.Create(db, pp)
srv.set_keyset_capacity(2 * NCL)
N, mods = enc.poly_modulus_degree, enc.coeff_modulus
k = len(mods) - 1
pid_k, pid_q = W.parms_id(N, mods, enc.plain_modulus), W.parms_id(N, mods[:-1], enc.plain_modulus)
reqs = []
for c in range(NCL):
    ck = {}
    for g, key in keys.items():
        kk = key.copy()
        for i in range(k + 1):
            kk[:, :, i, :] = (kk[:, :, i, :] + np.uint64(1000 + c)) % np.uint64(mods[i])
        ck[g] = kk
    reqs.append(W.save_request([queries[c]], W.save_galois_keys(ck, N, pid_k), pid_q))
for _ in range(3):
    srv.ProcessRequests(reqs)
srv.sync()
time.sleep(0.2)
t0 = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
w0 = time.perf_counter()
for _ in range(calls):
    out = srv.ProcessRequests(reqs)
w1 = time.perf_counter()
t1 = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
assert all(s == 0 for s, _ in out)
print("WINDOW %d %d calls %d queries_per_s %.1f" % (t0, t1, calls, calls * NCL / (w1 - w0)))
