"""The slot-sharded multi-GPU step (pirgpu_slots_*, pir_amd.distributed.run_batch_slots / SlotsPipeline) with the REAL
server on one GPU.

Every rank of the step holds the NTT slots [cut[g], cut[g+1]) of EVERY plaintext (the base case of
PIRDatabase::multiply, reference database.cpp:185-194, is a dyadic product in NTT form: independent per slot), receives
its slots of every query's packed column selectors, scans them against full rows and returns the row sums to the rank
that expanded the query, which runs the upper level (database.cpp:196-254) itself.  Here G slot-shard contexts live on
cuda:0 and the two all-to-alls are played by tensor copies (in-process) or by gloo (two processes): every byte the step
computes is compared with the oracle's full-database reply (reference server.cpp:173-195)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _setup(items, elem=288, N=4096, plain_bits=24, moduli=None, t=None):
    import pir_amd
    from gpu_helpers import to_product_params
    from pir_fixtures import PirSetup
    s = PirSetup(items, elem, 2, N=N, plain_bits=plain_bits, moduli=moduli, t=t)
    return s, to_product_params(s.params), pir_amd


def _rank_server(pir_amd, pp, s, slots):
    db = pir_amd.PIRDatabase.Create(pp, s.raw, slots=slots)
    db.finalize(release_staging=True)          # a slot shard keeps only its 1 / G of the operand layout
    srv = pir_amd.PIRServer(db, pp)
    srv.set_galois_keys(s.galois_keys)
    srv.set_concurrency(16)
    return srv


from gpu_helpers import all_to_all_in_process as _all_to_all   # noqa: E402


@pytest.mark.parametrize("G,items,per", [(2, 3000, 4), (8, 3000, 1), (4, 12000, 9), (8, 40000, 2), (3, 3000, 3), (8, 3000, 17)])
def test_slot_shards_on_one_gpu_reproduce_the_oracle(G, items, per):
    """G slot-shard contexts, `per` queries each (9: a full and a partial group per rank; G = 3: uneven slot cuts; 17 on
    8 ranks: 24 groups per scan = two launches of at most 16); 12000 / 40000 items: 18 x 18 and 32 x 32 matrices (two row
    tiles, ragged; two column groups)."""
    from pir_amd import distributed as D
    s, pp, pir_amd = _setup(items)
    p = s.params
    kN = s.orc.k * 4096
    cuts = D.slot_cuts(kN, G)
    srvs = [_rank_server(pir_amd, pp, s, (cuts[g], cuts[g + 1])) for g in range(G)]
    assert all(D.slots_exchange_supported(v) for v in srvs)
    batch = G * per
    indexes = [(items - 1 - 131 * i) % items for i in range(batch)]
    queries = np.stack([s.client.create_query_for(p, i) for i in indexes])
    bufs = [D.SlotsBuffers(srvs[g], batch, g, G, torch, "cuda:0") for g in range(G)]
    for rep in range(2):                        # twice: buffers, lanes and workers are reused
        for g in range(G):
            # second round: the row sums put together by a separate assembly pass instead of inside the inverse transform
            srvs[g].db.set_option("slots_gather_ntt", 1 - rep)
            srvs[g].db.set_option("slots_scan_blk_major", rep)      # ... and the scan's units in (slot block, group) order
            srvs[g].stage_batch(queries)
            srvs[g].slots_expand_async(g * per, per, bufs[g].packed_send.data_ptr(), bufs[g].sv.data_ptr(), cuts)
            srvs[g].sync()
        _all_to_all([b.packed_recv for b in bufs], [b.packed_send for b in bufs], [b.x1_recv for b in bufs],
                    [b.x1_send for b in bufs])
        for g in range(G):
            srvs[g].slots_scan_async(bufs[g].packed_recv.data_ptr(), G, per, bufs[g].rows_send.data_ptr())
            srvs[g].sync()
        _all_to_all([b.rows_recv for b in bufs], [b.rows_send for b in bufs], [b.x2_recv for b in bufs],
                    [b.x2_send for b in bufs])
        for g in range(G):
            srvs[g].slots_finish_async(bufs[g].rows_recv.data_ptr(), per, bufs[g].sv.data_ptr(), cuts,
                                       bufs[g].replies.data_ptr())
            srvs[g].sync()
        for g in range(G):
            mine = bufs[g].replies.cpu().numpy().view(np.uint64)
            for i in range(per):
                if per > 9 and (g * per + i) % 7 and rep:        # the big batch: every query once, a sample twice
                    continue
                rc, want = s.orc.process_query(s.db_ntt, p.dimensions, queries[g * per + i], s.galois_keys)
                assert rc == 0
                assert np.array_equal(mine[i], want), (rep, g, i)
                if rep == 0 and i == 0:
                    assert s.client.process_response(p, indexes[g * per + i], mine[i]) == s.item(indexes[g * per + i])
    # a slot shard holds 1 / G of every plaintext: the plain entry points refuse it
    from pir_amd.server import PirGpuError
    with pytest.raises(PirGpuError) as e:
        srvs[0].process_query(queries[0])
    assert e.value.code == 9 and "slot shard" in e.value.message
    for v in srvs:
        v.db.close()


def test_whole_context_serves_the_slots_step_alone_and_pipelined():
    """World size 1: a context that holds all slots runs the step through run_batch_slots and SlotsPipeline (Comm
    copies); four pipelined steps over different queries, three buffer sets."""
    from pir_amd import distributed as D
    items, batch, steps = 3000, 3, 5
    s, pp, pir_amd = _setup(items)
    p = s.params
    srv = _rank_server(pir_amd, pp, s, None)
    idx = [(items - 3 - 97 * i) % items for i in range(batch * steps)]
    q_all = np.stack([s.client.create_query_for(p, i) for i in idx])
    srv.stage_batch(q_all)
    want = [s.orc.process_query(s.db_ntt, p.dimensions, q, s.galois_keys)[1] for q in q_all]
    bufs = D.SlotsBuffers(srv, batch, 0, 1, torch, "cuda:0")
    ph = D.run_batch_slots(srv, bufs, None, 0, 1, D.Comm(None, 1), first=batch)
    assert set(ph) == {"expand_ms", "exchange_selectors_ms", "scan_ms", "exchange_rowsums_ms", "finish_ms"}
    got = bufs.replies.cpu().numpy().view(np.uint64)
    for i in range(batch):
        assert np.array_equal(got[i], want[batch + i])
    pipe = D.SlotsPipeline(srv, batch, 0, 1, None, torch, "cuda:0", comm=D.Comm(None, 1, host_sync=False))
    seen = {}
    for t in range(steps):
        pipe.submit(first=t * batch)
        if t >= 2:
            pipe.streams.fin.synchronize()
            seen[t - 2] = pipe.replies(t - 2).cpu().numpy().view(np.uint64).copy()
    pipe.flush()
    for t in (steps - 2, steps - 1):
        seen[t] = pipe.replies(t).cpu().numpy().view(np.uint64).copy()
    for t in range(steps):
        for i in range(batch):
            assert np.array_equal(seen[t][i], want[t * batch + i]), (t, i)
    # the plain pipeline of the same context gives the same replies
    assert np.array_equal(srv.process_batch(q_all[:batch]), np.stack(want[:batch]))
    srv.db.close()


def test_synchronous_slots_step_with_five_byte_row_sums(monkeypatch):
    """run_batch_slots with PIRGPU_SLOTS_PACK40=1: the pack kernel runs on the library's stream, the exchange on torch's
    current one -- the step orders them with host waits (an unordered pack -> all-to-all edge gave the
    collective a send buffer that had not been written yet).  Several repetitions over different queries with the send
    buffer poisoned in between, so that a collective that ran ahead of the pack could not pass."""
    from pir_amd import distributed as D
    monkeypatch.setenv("PIRGPU_SLOTS_PACK40", "1")
    items, batch, steps = 3000, 8, 4
    s, pp, pir_amd = _setup(items)
    p = s.params
    srv = _rank_server(pir_amd, pp, s, None)
    assert srv.pack40_supported()
    idx = [(items - 5 - 89 * i) % items for i in range(batch * steps)]
    q_all = np.stack([s.client.create_query_for(p, i) for i in idx])
    srv.stage_batch(q_all)
    bufs = D.SlotsBuffers(srv, batch, 0, 1, torch, "cuda:0")
    assert bufs.rows40 and bufs.rc == 2 * srv.scan_info()["rows"]
    for t in range(steps):
        bufs.rows_send40.fill_(-1)
        bufs.rows_recv40.fill_(-1)
        bufs.rows_recv.fill_(-1)
        D.run_batch_slots(srv, bufs, None, 0, 1, D.Comm(None, 1, host_sync=(t % 2 == 0)), first=t * batch)
        got = bufs.replies.cpu().numpy().view(np.uint64)
        for i in range(0, batch, 3):
            rc, want = s.orc.process_query(s.db_ntt, p.dimensions, q_all[t * batch + i], s.galois_keys)
            assert rc == 0 and np.array_equal(got[i], want), (t, i)
    srv.db.close()


def test_slot_range_errors():
    from pir_amd.server import PirGpuError
    s, pp, pir_amd = _setup(3000)
    for bad in ((8, 1024), (0, 1000), (1024, 512), (0, 2 * 8192)):
        with pytest.raises(PirGpuError) as e:
            pir_amd.PIRDatabase.Create(pp, slots=bad)
        assert e.value.code == 3
    with pytest.raises(PirGpuError):                      # a slot shard keeps all rows
        pir_amd.PIRDatabase.Create(pp, slots=(0, 1024), shard=(0, 4))
    from pir_fixtures import PirSetup
    from gpu_helpers import to_product_params
    s1 = PirSetup(120, 288, 1, N=4096, plain_bits=24)      # d = 1: rows do not exist
    with pytest.raises(PirGpuError):
        pir_amd.PIRDatabase.Create(to_product_params(s1.params), slots=(0, 1024))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank(rank, world, port, items, per, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    try:
        from pir_amd import distributed as D
        s, pp, pir_amd = _setup(items)
        p = s.params
        cuts = D.slot_cuts(s.orc.k * 4096, world)
        srv = _rank_server(pir_amd, pp, s, (cuts[rank], cuts[rank + 1]))
        batch, steps = per * world, 4
        idx = [(items - 5 - 61 * i) % items for i in range(batch * steps)]
        q_all = np.stack([s.client.create_query_for(p, i) for i in idx])
        srv.stage_batch(q_all)
        comm = D.Comm(dist, world)
        bufs = D.SlotsBuffers(srv, batch, rank, world, torch, "cuda:0")
        D.run_batch_slots(srv, bufs, dist, rank, world, comm)
        mine = bufs.replies.cpu().numpy().view(np.uint64)
        for i in range(per):
            want = s.orc.process_query(s.db_ntt, p.dimensions, q_all[rank * per + i], s.galois_keys)[1]
            ok &= bool(np.array_equal(mine[i], want))
        pipe = D.SlotsPipeline(srv, batch, rank, world, dist, torch, "cuda:0", comm=D.Comm(dist, world, host_sync=False))
        for t in range(steps):
            pipe.submit(first=t * batch)
        pipe.flush()
        for t in (steps - 3, steps - 2, steps - 1):        # three buffer sets: the last three steps are still there
            got = pipe.replies(t).cpu().numpy().view(np.uint64)
            for i in range(per):
                g = t * batch + rank * per + i
                want = s.orc.process_query(s.db_ntt, p.dimensions, q_all[g], s.galois_keys)[1]
                same = bool(np.array_equal(got[i], want))
                if not same:
                    print("rank %d: pipelined slots step %d query %d differs" % (rank, t, i), flush=True)
                ok &= same
        srv.db.close()
    finally:
        out_q.put((rank, ok))
        dist.destroy_process_group()


@pytest.mark.parametrize("world,per", [(2, 9), (8, 2)])
def test_slots_step_with_real_servers_over_gloo(world, per):
    """`world` processes share the GPU, each with its slot-shard context of the real server; the all-to-alls run over
    gloo through host memory (Comm) -- every line of the multi-GPU step except the RCCL calls themselves."""
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, 3000, per, out_q)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(600)
        assert pr.exitcode == 0
    got = dict(out_q.get(timeout=5) for _ in range(world))
    assert got == {r: True for r in range(world)}
