#!/usr/bin/env python3
"""Sustained load on the wire boundary: N caller threads, each sending its 64 clients' serialized requests through
pirgpu_process_requests in a loop (what bench.py reports as wire_multi_client_qps), alone in a process so that a kernel
trace of it shows only this.  usage: r04_wire_load.py [callers] [calls_per_caller] [capacity]"""
import ctypes as C, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pir_amd

callers = int(sys.argv[1]) if len(sys.argv) > 1 else 2
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cap = int(sys.argv[3]) if len(sys.argv) > 3 else 128
n_w = 64
enc = pir_amd.generate_encryption_params(4096, 24)
pp = pir_amd.create_pir_parameters(1 << 20, 288, 2, enc)
raw = np.random.default_rng(42).integers(0, 256, size=(pp.num_items, pp.bytes_per_item), dtype=np.uint8)
db = pir_amd.PIRDatabase.Create(pp)
db.populate(raw)
db.finalize(release_staging=True)
srv = pir_amd.PIRServer.Create(db, pp)
srv.set_keyset_capacity(cap)
wcl = [pir_amd.PIRClient.Create(pp, seed=b"load-%d" % i) for i in range(n_w)]
reqs = [c.CreateRequest([(7919 * i + 13) % pp.num_items]) for i, c in enumerate(wcl)]
first = srv.ProcessRequests(reqs)
assert all(st == 0 for st, _ in first)
lib, handle = srv.lib, srv.db.handle
views = [np.frombuffer(r, dtype=np.uint8) for r in reqs]
ptrs = (C.c_void_p * n_w)(*[b.ctypes.data for b in views])
lens = (C.c_size_t * n_w)(*[len(r) for r in reqs])

def one(st):
    resp, rlen, status = st
    lib.pirgpu_process_requests(handle, n_w, ptrs, lens, resp, rlen, status)
    ok = all(status[i] == 0 for i in range(n_w))
    for i in range(n_w):
        if status[i] == 0:
            lib.pirgpu_free(resp[i])
    return ok

gate = threading.Barrier(callers + 1)
oks = [True] * callers
def caller(ci):
    st = ((C.c_void_p * n_w)(), (C.c_size_t * n_w)(), (C.c_int * n_w)())
    one(st); one(st)
    gate.wait()
    for _ in range(calls):
        oks[ci] = one(st) and oks[ci]
    gate.wait()
ths = [threading.Thread(target=caller, args=(i,)) for i in range(callers)]
for t in ths: t.start()
gate.wait(); t0 = time.perf_counter(); gate.wait(); dt = time.perf_counter() - t0
for t in ths: t.join()
print("callers %d calls %d capacity %d: %.1f queries/s, %.3f ms per window of 64, all ok %s" %
      (callers, calls, cap, callers * calls * n_w / dt, dt / (callers * calls) * 1e3, all(oks)), flush=True)
