"""Host-side phase times of a window of 16 clients' requests (PIRGPU_WIRE_TRACE=1)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import pir_amd, bench
import seal_wire as W
class A: pass
args = A(); args.config = 3; args.log_items = 20; args.dims = 2
enc, pp, _ = bench.build_workload(args, pir_amd)
NCL = int(sys.argv[1]) if len(sys.argv) > 1 else 16
raw, keys, queries = bench.synthetic_inputs(pp, n_queries=NCL)
db = pir_amd.PIRDatabase.Create(pp); db.populate(raw); db.finalize(release_staging=True)
srv = pir_amd.PIRServer.Create(db, pp)
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
os.environ["PIRGPU_WIRE_TRACE"] = "1"
import ctypes as C
lib, handle = srv.lib, db.handle
n = len(reqs)
bufs = [np.frombuffer(r, dtype=np.uint8) for r in reqs]
ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
lens = (C.c_size_t * n)(*[len(r) for r in reqs])
resp, rlen, status = (C.c_void_p * n)(), (C.c_size_t * n)(), (C.c_int * n)()
t0 = time.perf_counter(); lib.pirgpu_process_requests(handle, n, ptrs, lens, resp, rlen, status); t1 = time.perf_counter()
for i in range(n): lib.pirgpu_free(resp[i])
t2 = time.perf_counter()
print("C call ms", (t1 - t0) * 1e3, "frees ms", (t2 - t1) * 1e3)
