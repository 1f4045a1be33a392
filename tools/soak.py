#!/usr/bin/env python3
"""Randomised soak of the batch pipeline on one GPU: random matrix shapes, batch sizes, worker counts and
repeat counts; every batch reply must equal the single-query reply of the same query, and one reply per
configuration is checked against the CPU oracle.  Round 3: every configuration also serves a random number of
CLIENTS with different Galois keys -- every query of a batch is assigned a random client's resident key set, so the
groups of 8 mix clients (sometimes with fewer slots than clients: evictions) -- and replies are checked against the
oracle run with that client's keys.  Round 5: half of the d = 2 configurations also run the SLOT-sharded multi-GPU step
(1 - 8 slot-shard contexts on the one GPU, all-to-alls as tensor copies) on the same queries and key sets; its replies must
equal the plain pipeline's.  Run on the GPU box:  python tools/soak.py [seconds]"""
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
    n_cfg = n_batches = n_queries = n_slots_steps = n_slots_queries = 0
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
            # round 5: the SLOT-sharded step on the same queries -- a random number of slot-shard contexts (each holding
            # 1 / G of the NTT slots of every plaintext), a random number of queries per "rank", mixed clients inside the
            # groups; the all-to-alls are tensor copies; every reply must equal the plain batch pipeline's
            if info["mfma"] and info["chunks"] == 1 and d == 2 and rng.integers(0, 2):
                from pir_amd import distributed as D
                from gpu_helpers import all_to_all_in_process
                G = int(rng.choice([1, 2, 3, 4, 8]))
                per = count // G
                if per >= 1:
                    kN = s.orc.k * 4096
                    cuts = D.slot_cuts(kN, G)
                    ranks = []
                    for g in range(G):
                        dbg = pir_amd.PIRDatabase.Create(pp, s.raw, slots=(cuts[g], cuts[g + 1]) if G > 1 else None)
                        if rng.integers(0, 2):
                            dbg.finalize(release_staging=True)
                        sg = pir_amd.PIRServer(dbg, pp)
                        sg.set_galois_keys(s.galois_keys)
                        sg.set_concurrency(int(rng.choice([8, 9, 16])))
                        mine = who[g * per:(g + 1) * per]
                        sl = {w: sg.install_keyset(b"soak-client-%d" % w, ckeys[w]) for w in sorted(set(mine))}
                        sg.stage_batch(queries[:G * per])
                        sg.set_batch_keysets([sl[w] if g * per <= i < (g + 1) * per else 0 for i, w in enumerate(who[:G * per])])
                        ranks.append((dbg, sg, D.SlotsBuffers(sg, G * per, g, G, torch, "cuda:0")))
                    for g, (dbg, sg, b) in enumerate(ranks):
                        sg.slots_expand_async(g * per, per, b.packed_send.data_ptr(), b.sv.data_ptr(), cuts)
                        sg.sync()
                    bufs = [r[2] for r in ranks]
                    all_to_all_in_process([b.packed_recv for b in bufs], [b.packed_send for b in bufs],
                                          [b.x1_recv for b in bufs], [b.x1_send for b in bufs])
                    for dbg, sg, b in ranks:
                        sg.slots_scan_async(b.packed_recv.data_ptr(), G, per, b.rows_send.data_ptr())
                        sg.sync()
                    all_to_all_in_process([b.rows_recv for b in bufs], [b.rows_send for b in bufs],
                                          [b.x2_recv for b in bufs], [b.x2_send for b in bufs])
                    for g, (dbg, sg, b) in enumerate(ranks):
                        sg.slots_finish_async(b.rows_recv.data_ptr(), per, b.sv.data_ptr(), cuts, b.replies.data_ptr())
                        sg.sync()
                        mine_r = b.replies.cpu().numpy().view(np.uint64)
                        for i in range(per):
                            assert np.array_equal(mine_r[i], got[g * per + i]), ("slots != plain", dims, G, per, g, i, who)
                        dbg.close()
                    n_slots_steps += 1
                    n_slots_queries += G * per
        db.close()
        print("cfg %d dims=%s mfma=%s ok (%d batches, %d queries so far)" % (n_cfg, dims, info["mfma"], n_batches, n_queries),
              flush=True)
    print("soak OK: %d configurations, %d batches, %d queries; %d slot-sharded steps, %d queries"
          % (n_cfg, n_batches, n_queries, n_slots_steps, n_slots_queries))


if __name__ == "__main__":
    main()
