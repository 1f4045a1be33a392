"""cfg 5 at full size: the slot-sharded step (two slot-shard contexts, one after the other) against the plain pipeline,
with the scan's unit order and the finish's gather switched per run -- which of them breaks rank 1's reply?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle, pir_amd
from pir_amd import distributed as D
from gpu_helpers import random_ct, random_key, to_product_params, all_to_all_in_process

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
if cfg == 5:
    N = 16384; m = oracle.BFV_DEFAULT[N]; moduli = m[:4] + [m[8]]; n_items, item_bytes = 1 << 24, 288
else:
    N = 8192; m = oracle.BFV_DEFAULT[N]; moduli = m[:3] + [m[4]]; n_items, item_bytes = 1 << 22, 1024
params = oracle.create_pir_parameters(n_items, item_bytes, 2, N=N, moduli=moduli, plain_bits=24)
orc = oracle.Oracle.from_params(params)
rng = np.random.default_rng(59)
raw = rng.integers(0, 256, size=(n_items, item_bytes), dtype=np.uint8)
keys = {(N >> j) + 1: random_key(orc, rng) for j in range(N.bit_length() - 1)}
NQ = int(os.environ.get("NQ", "2"))
queries = random_ct(orc, rng, NQ)[:, None]
pp = to_product_params(params)
db = pir_amd.PIRDatabase.Create(pp, raw); db.finalize(release_staging=True)
srv = pir_amd.PIRServer.Create(db, pp); srv.set_galois_keys(keys)
batch = srv.process_batch(queries, n_workers=min(NQ, 8))
db.close()
G = 2
per = NQ // G
cuts = D.slot_cuts(orc.k * N, G)
for variant in [dict(), dict(slots_scan_blk_major=0), dict(slots_gather_ntt=0), dict(slots_scan_blk_major=0, slots_gather_ntt=0)]:
    bufs = None
    for g in range(G):
        dbg = pir_amd.PIRDatabase.Create(pp, raw, slots=(cuts[g], cuts[g + 1])); dbg.finalize(release_staging=True)
        for k_, v_ in variant.items(): dbg.set_option(k_, v_)
        sg = pir_amd.PIRServer(dbg, pp); sg.set_galois_keys(keys); sg.set_concurrency(8)
        if g == 0:
            bufs = [D.SlotsBuffers(sg, NQ, r, G, torch, "cuda:0") for r in range(G)]
            sg.stage_batch(queries)
            for r in range(G):
                sg.slots_expand_async(r * per, per, bufs[r].packed_send.data_ptr(), bufs[r].sv.data_ptr(), cuts)
            sg.sync()
            all_to_all_in_process([b.packed_recv for b in bufs], [b.packed_send for b in bufs], [b.x1_recv for b in bufs], [b.x1_send for b in bufs])
        sg.slots_scan_async(bufs[g].packed_recv.data_ptr(), G, per, bufs[g].rows_send.data_ptr()); sg.sync()
        if g == G - 1:
            all_to_all_in_process([b.rows_recv for b in bufs], [b.rows_send for b in bufs], [b.x2_recv for b in bufs], [b.x2_send for b in bufs])
            for r in range(G):
                sg.slots_finish_async(bufs[r].rows_recv.data_ptr(), per, bufs[r].sv.data_ptr(), cuts, bufs[r].replies.data_ptr())
            sg.sync()
        dbg.close()
    res = []
    for r in range(G):
        got = bufs[r].replies.cpu().numpy().view(np.uint64)
        for i in range(per):
            same = np.array_equal(got[i], batch[r * per + i])
            if not same:
                diff = np.argwhere(got[i] != batch[r * per + i])
                res.append((r, i, "DIFF n=%d first=%s" % (len(diff), diff[0].tolist())))
            else:
                res.append((r, i, "ok"))
    print("cfg", cfg, "variant", variant, res, flush=True)
