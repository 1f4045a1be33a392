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

SLOTS = "--slots" in sys.argv
if SLOTS:
    sys.argv.remove("--slots")
class A: pass
args = A(); args.config = int(sys.argv[1]) if len(sys.argv) > 1 else 3; args.log_items = 20; args.dims = 2
enc, pp, _ = bench.build_workload(args, pir_amd)
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64     # queries the group of G shards serves per step (hybrid: 64 / R)
raw, keys, queries = bench.synthetic_inputs(pp, n_queries=batch)

def timed(fn, srv, n, warm=2):
    for _ in range(warm): fn()
    srv.sync(); t0 = time.perf_counter()
    for _ in range(n): fn()
    srv.sync()
    return (time.perf_counter() - t0) / n * 1e3

def slots_budget():
    """--slots: the slot-sharded step (pirgpu_slots_*).  A context holding the first 1 / G of the slots of every plaintext
    runs what one rank of a G-GPU job runs per step of `batch` queries: E for its batch / G queries, S for all of them
    (the other ranks' packed slices are copies of its own), U for its own; each alone and queued in the pipeline's order
    (S_(s-1), E_s, U_(s-2) per step), next to the plain single-GPU step measured in the same process."""
    out = {}
    nn = 20 if args.config == 3 else 4
    plain_db = pir_amd.PIRDatabase.Create(pp, device=0); plain_db.populate(raw); plain_db.finalize(release_staging=True)
    plain = pir_amd.PIRServer(plain_db, pp); plain.set_galois_keys(keys); plain.set_concurrency(16)
    plain.stage_batch(queries)
    out["plain_single_gpu_step_ms"] = round(timed(plain.run_batch, plain, nn), 3)
    print("plain", out["plain_single_gpu_step_ms"], flush=True)
    plain_db.close()
    kN = (len(pp.encryption_parameters.coeff_modulus) - 1) * pp.encryption_parameters.poly_modulus_degree
    for G in ([int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (1, 2, 4, 8)):
        cuts = D.slot_cuts(kN, G)
        db = pir_amd.PIRDatabase.Create(pp, device=0, slots=(cuts[0], cuts[1]) if G > 1 else None)
        db.populate(raw); db.finalize(release_staging=True)
        srv = pir_amd.PIRServer(db, pp); srv.set_galois_keys(keys); srv.set_concurrency(16)
        srv.stage_batch(queries)
        # three buffer sets as in SlotsPipeline (one where three would not fit next to the lanes: cfg 5 at G = 2)
        n_sets = 3 if args.config != 5 or G >= 4 else 1
        bufs = [D.SlotsBuffers(srv, batch, 0, G, torch, "cuda:0") for _ in range(n_sets)]
        bufs = (bufs * 3)[:3]
        per = bufs[0].per
        def E(b=bufs[0]):
            srv.slots_expand_async(0, per, b.packed_send.data_ptr(), b.sv.data_ptr(), b.cuts)
        def S(b=bufs[0]):
            srv.slots_scan_async(b.packed_recv.data_ptr(), G, per, b.rows_send.data_ptr())
        def U(b=bufs[0]):
            srv.slots_finish_async(b.rows_recv.data_ptr(), per, b.sv.data_ptr(), b.cuts, b.replies.data_ptr())
        for b in bufs:
            E(b); srv.sync()
            piece = b.groups * b.piece[0]
            for r in range(G):      # every source rank's piece of my slots: copies of my own first piece
                b.packed_recv[r * piece:(r + 1) * piece].copy_(b.packed_send[:piece])
            b.rows_recv.zero_()
        torch.cuda.synchronize()
        res = {"E_ms": timed(E, srv, nn), "S_ms": timed(S, srv, nn), "U_ms": timed(U, srv, nn)}
        step = [0]
        def pipelined():
            s_ = step[0]; step[0] += 1
            S(bufs[(s_ + 2) % 3]); E(bufs[s_ % 3]); U(bufs[(s_ + 1) % 3])
        res["S_E_U_queued_together_ms"] = timed(pipelined, srv, nn, warm=3)
        # other orders of the three calls of a step (the lanes are handed out round robin, call by call): U first puts
        # [U, S] on one lane and E on the other -- the expansion's narrow, latency-bound first levels then run beside the
        # VALU-bound upper level, its wide levels beside the HBM-bound scan
        def order(seq):
            def f():
                s_ = step[0]; step[0] += 1
                for ch in seq:
                    {"S": S, "E": E, "U": U}[ch](bufs[(s_ + "ESU".index(ch) * 2) % 3])
            return f
        for seq in ("UES", "USE", "EUS", "ESU", "SUE"):
            res["queued_together_order_%s_ms" % seq] = timed(order(seq), srv, nn, warm=3)
        for wgs in (0, 64, 192):
            db.set_option("slots_scan_wgs", wgs)
            res["queued_together_scan_wgs_%d_ms" % wgs] = timed(pipelined, srv, nn, warm=3)
        db.set_option("slots_scan_wgs", 128)
        # the scan's units ordered (slot block, group): neighbouring workgroups read the same database tiles
        db.set_option("slots_scan_blk_major", 1)
        res["S_blk_major_ms"] = timed(S, srv, nn)
        res["queued_together_blk_major_ms"] = timed(pipelined, srv, nn, warm=3)
        db.set_option("slots_scan_wgs", 0)
        res["S_blk_major_all_cus_ms"] = timed(S, srv, nn)
        db.set_option("slots_scan_blk_major", 0)
        res["S_all_cus_ms"] = timed(S, srv, nn)
        db.set_option("slots_scan_wgs", 128)
        db.set_option("slots_gather_ntt", 0)
        res["U_separate_assembly_ms"] = timed(U, srv, nn)
        db.set_option("slots_gather_ntt", 1)
        res["recv_MB_per_step"] = bufs[0].exchange_bytes_per_query(G) * batch / 1e6
        res["recv_selectors_MB"] = (G - 1) * bufs[0].groups * bufs[0].piece[0] / 1e6
        res["recv_rowsums_MB_u64"] = per * bufs[0].rc * (kN - bufs[0].mine) * 8 / 1e6
        res["db_slice_MB"] = srv.scan_bytes() / 1e6
        res["speedup_vs_plain_if_links_hide"] = out["plain_single_gpu_step_ms"] / res["S_E_U_queued_together_ms"]
        out["G=%d" % G] = {k: round(v, 3) for k, v in res.items()}
        print("G=%d" % G, out["G=%d" % G], flush=True)
        db.close()
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "rank_budget_slots_cfg%d%s.json" % (args.config, "" if batch == 64 else "_q%d" % batch)), "w"), indent=1)

if SLOTS:
    slots_budget()
    sys.exit(0)

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
