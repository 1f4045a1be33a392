"""Localises the N = 8192, 42-bit t, d = 2 mismatch (tests/test_gpu_large_rings.py::test_correctness_test_n8192_t42_d2)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pir_amd
from gpu_helpers import to_product_params
from pir_fixtures import PirSetup

def run(tag, opts=None, env=None, bits=42, dbsize=87, d=2):
    for k, v in (env or {}).items():
        os.environ[k] = v
    try:
        s = PirSetup(dbsize, 0, d, N=8192, plain_bits=bits)
        p = s.params
        pp = to_product_params(p)
        db = pir_amd.PIRDatabase.Create(pp)
        for k, v in (opts or {}).items():
            db.set_option(k, v)
        db.populate(s.raw)
        srv = pir_amd.PIRServer.Create(db, pp)
        srv.set_galois_keys(s.galois_keys)
        q = s.client.create_query_for(p, 5)
        rc, sv = s.orc.oblivious_expansion_multi(q, p.dim_sum, s.galois_keys)
        rc, exp = s.orc.db_multiply(s.db_ntt, p.dimensions, sv.copy())
        got_m = db.multiply(sv)
        got_q = srv.process_query(q)
        pts_ok = all(np.array_equal(db.read_plaintext(i), s.db_ntt[i]) for i in range(p.num_pt))
        bad_m = [int(i) for i in range(exp.shape[0]) if not np.array_equal(got_m[i], exp[i])]
        bad_q = [int(i) for i in range(exp.shape[0]) if not np.array_equal(got_q[i], exp[i])]
        print(tag, "dims", p.dimensions, "ER", db.expansion_ratio(), "scan", srv.scan_info(), "db_ok", pts_ok,
              "multiply bad cts", bad_m, "query bad cts", bad_q, flush=True)
        if bad_m:
            i = bad_m[0]
            diff = np.argwhere(got_m[i] != exp[i])
            print("   first diffs (poly, residue, coeff):", diff[:5].tolist(), "count", len(diff), flush=True)
        db.close()
    finally:
        for k in (env or {}):
            del os.environ[k]

run("default")
run("t=24", bits=24)
run("t=32", bits=32)
run("t=33", bits=33)
run("scan_mfma=0", {"scan_mfma": 0})
run("NTT_MODE=0", env={"PIRGPU_NTT_MODE": "0"})
run("NTT_MODE=2", env={"PIRGPU_NTT_MODE": "2"})
run("split_upper", {"split_upper": 1})
run("d=1", d=1)
