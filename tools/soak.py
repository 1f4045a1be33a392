#!/usr/bin/env python3
"""Randomised soak of the batch pipeline on one GPU: random matrix shapes, batch sizes, worker counts and
repeat counts; every batch reply must equal the single-query reply of the same query, and one reply per
configuration is checked against the CPU oracle.  Run on the GPU box:  python tools/soak.py [seconds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (torch's HIP runtime first)
import oracle  # noqa: E402
import pir_amd  # noqa: E402
from gpu_helpers import to_product_params  # noqa: E402
from pir_fixtures import PirSetup  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "1")))
    t_end = time.time() + budget
    n_cfg = n_batches = n_queries = 0
    while time.time() < t_end:
        d = int(rng.choice([1, 2, 2, 2, 3]))
        if d == 1:
            dims = [int(rng.integers(1, 40))]
        elif d == 2:
            dims = [int(rng.integers(1, 40)), int(rng.integers(1, 70))]
        else:
            dims = [int(rng.integers(1, 6)), int(rng.integers(1, 6)), int(rng.integers(1, 30))]
        pts = int(np.prod(dims))
        short = int(rng.integers(0, 3))
        probe = oracle.create_pir_parameters(10, 2048, 1, N=4096, plain_bits=24)
        dbsize = max(1, pts * probe.items_per_plaintext - short)
        s = PirSetup(dbsize, 2048, d, N=4096, plain_bits=24, seed=int(rng.integers(1 << 30)))
        if s.params.num_pt != pts:
            continue
        s.params.dimensions = dims
        pp = to_product_params(s.params)
        db = pir_amd.PIRDatabase.Create(pp, s.raw)
        srv = pir_amd.PIRServer(db, pp)
        srv.set_galois_keys(s.galois_keys)
        info = srv.scan_info()
        n_cfg += 1
        for _ in range(int(rng.integers(1, 4))):
            count = int(rng.integers(1, 21))
            workers = int(rng.choice([1, 2, 3, 5, 8, 9, 16]))
            idx = [int(rng.integers(0, dbsize)) for _ in range(count)]
            queries = np.stack([s.client.create_query_for(s.params, i) for i in idx])
            got = srv.process_batch(queries, n_workers=workers)
            pick = int(rng.integers(0, count))
            rc, exp = s.orc.process_query(s.db_ntt, s.params.dimensions, queries[pick], s.galois_keys)
            assert rc == 0 and np.array_equal(got[pick], exp), ("oracle mismatch", dims, count, workers, pick)
            for i in range(count):
                assert np.array_equal(got[i], srv.process_query(queries[i])), ("batch != single", dims, count, workers, i)
            n_batches += 1
            n_queries += count
        db.close()
        print("cfg %d dims=%s mfma=%s ok (%d batches, %d queries so far)" % (n_cfg, dims, info["mfma"], n_batches, n_queries),
              flush=True)
    print("soak OK: %d configurations, %d batches, %d queries" % (n_cfg, n_batches, n_queries))


if __name__ == "__main__":
    main()
