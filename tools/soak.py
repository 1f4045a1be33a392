#!/usr/bin/env python3
"""Randomised soak of the batch pipeline on one GPU: random matrix shapes, batch sizes, worker counts and
repeat counts; every batch reply must equal the single-query reply of the same query, and one reply per
configuration is checked against the CPU oracle.  Round 3: every configuration also serves a random number of
CLIENTS with different Galois keys -- every query of a batch is assigned a random client's resident key set, so the
groups of 8 mix clients (sometimes with fewer slots than clients: evictions) -- and replies are checked against the
oracle run with that client's keys.  Run on the GPU box:  python tools/soak.py [seconds]"""
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
from oracle.client import Client  # noqa: E402


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
        # clients: client 0 = the fixture's (its keys also sit in slot 0 through set_galois_keys)
        n_clients = int(rng.choice([1, 1, 2, 3, 5, 9]))
        clients = [s.client] + [Client(s.orc, seed=int(rng.integers(1 << 30))) for _ in range(n_clients - 1)]
        ckeys = [s.galois_keys] + [c.galois_keys() for c in clients[1:]]
        if n_clients > 2 and rng.integers(0, 2):
            srv.set_keyset_capacity(n_clients - 1)       # fewer slots than clients: re-installs evict
        for _ in range(int(rng.integers(1, 4))):
            count = int(rng.integers(1, 21))
            workers = int(rng.choice([1, 2, 3, 5, 8, 9, 16]))
            idx = [int(rng.integers(0, dbsize)) for _ in range(count)]
            who = [int(rng.integers(0, n_clients)) for _ in range(count)]
            queries = np.stack([clients[w].create_query_for(s.params, i) for w, i in zip(who, idx)])
            cap = srv.keyset_stats()["capacity"]
            if len(set(who)) > cap:                       # a batch can only name resident sets
                who = [w % cap for w in who]
                queries = np.stack([clients[w].create_query_for(s.params, i) for w, i in zip(who, idx)])
            slots = {w: srv.install_keyset(b"soak-client-%d" % w, ckeys[w]) for w in sorted(set(who))}
            srv.set_concurrency(workers)
            srv.stage_batch(queries)
            srv.set_batch_keysets([slots[w] for w in who])
            srv.run_batch()
            got = srv.fetch_batch()
            pick = int(rng.integers(0, count))
            rc, exp = s.orc.process_query(s.db_ntt, s.params.dimensions, queries[pick], ckeys[who[pick]])
            assert rc == 0 and np.array_equal(got[pick], exp), ("oracle mismatch", dims, count, workers, pick, who)
            for i in range(count):
                srv.use_keyset(slots[who[i]])
                assert np.array_equal(got[i], srv.process_query(queries[i])), ("batch != single", dims, count, workers, i, who)
            srv.use_keyset(0)
            n_batches += 1
            n_queries += count
        db.close()
        print("cfg %d dims=%s mfma=%s ok (%d batches, %d queries so far)" % (n_cfg, dims, info["mfma"], n_batches, n_queries),
              flush=True)
    print("soak OK: %d configurations, %d batches, %d queries" % (n_cfg, n_batches, n_queries))


if __name__ == "__main__":
    main()
