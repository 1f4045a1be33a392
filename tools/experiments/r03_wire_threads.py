"""Throughput of pirgpu_process_request under concurrent callers (T threads, each with its own client, calling in a
loop): requests that arrive while another is served are combined into windows by the leading thread."""
import os, sys, time, threading
import ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import pir_amd, bench
import seal_wire as W
class A: pass
args = A(); args.config = 3; args.log_items = 20; args.dims = 2
enc, pp, _ = bench.build_workload(args, pir_amd)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
raw, keys, queries = bench.synthetic_inputs(pp, n_queries=T)
db = pir_amd.PIRDatabase.Create(pp); db.populate(raw); db.finalize(release_staging=True)
srv = pir_amd.PIRServer.Create(db, pp)
N, mods = enc.poly_modulus_degree, enc.coeff_modulus
k = len(mods) - 1
pid_k, pid_q = W.parms_id(N, mods, enc.plain_modulus), W.parms_id(N, mods[:-1], enc.plain_modulus)
reqs = []
for c in range(T):
    ck = {}
    for g, key in keys.items():
        kk = key.copy()
        for i in range(k + 1):
            kk[:, :, i, :] = (kk[:, :, i, :] + np.uint64(1000 + c)) % np.uint64(mods[i])
        ck[g] = kk
    reqs.append(W.save_request([queries[c]], W.save_galois_keys(ck, N, pid_k), pid_q))
first = [srv.ProcessRequest(r) for r in reqs]          # installs every client's keys
lib, handle = srv.lib, db.handle
per_thread = int(sys.argv[2]) if len(sys.argv) > 2 else 40
errors = []
def work(t):
    buf = np.frombuffer(reqs[t], dtype=np.uint8)
    resp, rlen = C.c_void_p(), C.c_size_t()
    for it in range(per_thread):
        rc = lib.pirgpu_process_request(handle, buf.ctypes.data_as(C.POINTER(C.c_uint8)), len(reqs[t]), C.byref(resp), C.byref(rlen))
        if rc != 0:
            errors.append((t, rc)); return
        if it == per_thread - 1 and C.string_at(resp, rlen.value) != first[t]:
            errors.append((t, "differs"))
        lib.pirgpu_free(resp)
threads = [threading.Thread(target=work, args=(t,)) for t in range(T)]
t0 = time.perf_counter()
for th in threads: th.start()
for th in threads: th.join()
dt = time.perf_counter() - t0
print("threads", T, "requests", T * per_thread, "qps", round(T * per_thread / dt, 1), "errors", errors[:3])
