"""Per-rank compute budget of the row-sharded step at cfg 3, measured on ONE GPU: for G = 1, 2, 4, 8 a context holding the
first 1/G of the rows runs what one rank of a G-GPU job runs per step of 64 queries -- E: expansion + packing of its
64/G queries, M: scans + upper level of all 64 queries on its shard (the packed inputs of the other ranks are copies of
its own: contents do not matter for timing) -- and prints the times next to the bytes the rank would receive."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pir_amd
from pir_amd import distributed as D
import bench

class A: pass
args = A(); args.config = int(sys.argv[1]) if len(sys.argv) > 1 else 3; args.log_items = 20; args.dims = 2
enc, pp, _ = bench.build_workload(args, pir_amd)
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64     # queries the group of G shards serves per step (hybrid: 64 / R)
raw, keys, queries = bench.synthetic_inputs(pp, n_queries=batch)
out = {}
for G in ([int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (1, 2, 4, 8)):
    shard = D.shard_range(pp.dimensions[0], 0, G) if G > 1 else None
    db = pir_amd.PIRDatabase.Create(pp, device=0, shard=shard); db.populate(raw); db.finalize(release_staging=True)
    srv = pir_amd.PIRServer(db, pp); srv.set_galois_keys(keys); srv.set_concurrency(16)
    srv.stage_batch(queries)
    bufs = D.PackedBuffers(srv, batch, 0, G, torch, "cuda:0")
    per = bufs.per
    def E():
        srv.batch_expand_packed_async(0, per, bufs.packed[0].data_ptr(), bufs.rows_send.data_ptr(), bufs.cuts)
    def M():
        srv.batch_run_packed(bufs.packed.data_ptr(), G, per, bufs.rows_recv.data_ptr())
    E(); srv.sync()
    for r in range(1, G):
        bufs.packed[r].copy_(bufs.packed[0])
    bufs.rows_recv.zero_()
    torch.cuda.synchronize()
    res = {}
    for name, fn in (("E_ms", E), ("M_ms", M)):
        for _ in range(3): fn()
        srv.sync()
        t0 = time.perf_counter()
        n = 20 if args.config == 3 else 4
        for _ in range(n): fn()
        srv.sync()
        res[name] = (time.perf_counter() - t0) / n * 1e3
    def EM():
        E(); M()
    nn = 20 if args.config == 3 else 4
    for _ in range(2): EM()
    srv.sync(); t0 = time.perf_counter()
    for _ in range(nn): EM()
    srv.sync(); res["E_then_M_queued_together_ms"] = (time.perf_counter() - t0) / nn * 1e3
    # replicated-expansion form of the step (RowsReplicatedPipeline): the plain batch pipeline for ALL queries on the shard
    for _ in range(2): srv.run_batch()
    srv.sync(); t0 = time.perf_counter()
    for _ in range(nn): srv.run_batch()
    srv.sync(); res["replicated_form_step_ms"] = (time.perf_counter() - t0) / nn * 1e3
    res["recv_MB_per_step"] = bufs.exchange_bytes_per_query(G) * batch / 1e6
    res["packed_group_MB"] = bufs.sel_bytes / 1e6
    res["reduce_scatter_MB"] = (G - 1) / G * batch * db.reply_ct_count() * 2 * srv.k * srv.N * 8 / 1e6
    out["G=%d" % G] = {k: round(v, 3) for k, v in res.items()}
    print("G=%d" % G, out["G=%d" % G], flush=True)
    db.close()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "rank_budget_cfg%d%s.json" % (args.config, "" if batch == 64 else "_q%d" % batch)), "w"), indent=1)
