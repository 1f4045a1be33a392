"""World size EIGHT over gloo on the CPU: the shapes the driver's 8-GPU run of `bench.py --gpus 8` will hit first.

tests/test_distributed_gloo.py covers the glue at world size 2; at 8 ranks the headline configuration has row shards
of UNEVEN size (162 rows / 8 = 20 or 21 each), n0 not divisible by the world size, exactly one full group of 8
queries per rank (64 queries per step), and -- with other batch sizes -- partial groups.  The same irregularities at a
size the CPU oracle serves in seconds: a 21 x 21 matrix (shards of 3, 3, 3, 3, 3, 2, 2, 2 rows) and a 13 x 13 one whose
last ranks get ONE row, with per = 8 (one full group per rank), per = 3 (a partial group) and per = 9 (a full group + a
partial one).  Every rank drives the product glue (pir_amd.distributed: run_batch_rows_packed, RowsPipeline over three
steps, RowsReplicatedPipeline, the hybrid 2 x 4 layout) with the oracle-backed server of test_distributed_gloo.py and
checks the replies it ends up with against the oracle's full-database replies, bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from pir_amd import distributed as D   # noqa: E402

WORLD = 8


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def world8_check(rank, world, rows, per, mode):
    """One rank's part; True when every reply this rank owns equals the oracle's full-database reply."""
    from pir_fixtures import PirSetup
    from test_distributed_gloo import OracleShardServer
    n_pt = rows * rows - 2                       # ragged last row as well
    dbsize = n_pt * 40 - 7                       # 40 items of 288 bytes per plaintext at N = 4096, 24-bit t
    s = PirSetup(dbsize, 288, 2, N=4096, plain_bits=24)        # same seeds on every rank
    p = s.params
    assert list(p.dimensions) == [rows, rows], p.dimensions
    batch = per * world
    lo, hi = D.owned_queries(batch, rank, world)
    cuts = D.row_cuts(rows, world)
    sizes = [cuts[i + 1] - cuts[i] for i in range(world)]
    assert max(sizes) - min(sizes) == 1 and rows % world != 0     # the irregularity this test is about
    comm = D.Comm(dist, world)
    assert not comm.device_native
    D.check_sum_fits(max(s.orc.moduli[: s.orc.k]), world)
    ok = True

    def queries_for(step):
        idx = [(dbsize - 3 - 41 * (step * batch + i)) % dbsize for i in range(batch)]
        return idx, [s.client.create_query_for(p, i) for i in idx]

    def want(q):
        rc, rep = s.orc.process_query(s.db_ntt, p.dimensions, q, s.galois_keys)
        assert rc == 0
        return rep

    if mode == "packed":
        srv = OracleShardServer(s, rank, world)
        assert srv.hi - srv.lo == sizes[rank]
        idx, qs = queries_for(0)
        srv.stage_batch(qs)
        assert D.sync_zero_plaintexts(srv, dist, world, comm, torch, "cpu") == 0
        assert D.packed_exchange_supported(srv, dist, world, comm, torch, "cpu")
        bufs = D.PackedBuffers(srv, batch, rank, world, torch, "cpu")
        assert bufs.per == per
        D.run_batch_rows_packed(srv, bufs, dist, rank, world, comm)
        mine = bufs.replies.numpy().view(np.uint64)
        for i in range(lo, hi):                                 # rank r ends with the replies of ITS queries
            ok &= bool(np.array_equal(mine[i - lo], want(qs[i])))
            ok &= s.client.process_response(p, idx[i], mine[i - lo]) == s.item(idx[i])
    elif mode == "pipeline":
        # three consecutive steps over different queries: step t serves the staged queries [t * batch, (t + 1) * batch)
        srv = OracleShardServer(s, rank, world)
        steps = 3
        idx_all, q_all = [], []
        for t in range(steps):
            a, b = queries_for(t)
            idx_all += a
            q_all += b
        srv.stage_batch(q_all)
        D.sync_zero_plaintexts(srv, dist, world, comm, torch, "cpu")
        pipe = D.RowsPipeline(srv, batch, rank, world, dist, torch, "cpu", comm=D.Comm(dist, world))
        seen = {}
        for t in range(steps):
            pipe.submit(first=t * batch)
            if t >= 1:
                seen[t - 1] = pipe.replies(t - 1).numpy().view(np.uint64).copy()
        pipe.flush()
        seen[steps - 1] = pipe.replies(steps - 1).numpy().view(np.uint64).copy()
        for t in range(steps):
            for i in range(lo, hi):
                ok &= bool(np.array_equal(seen[t][i - lo], want(q_all[t * batch + i])))
    elif mode == "replicated":
        # every rank expands every query itself on its shard; the only collective is the reduce-scatter of the partials
        srv = OracleShardServer(s, rank, world)
        idx, qs = queries_for(0)
        srv.stage_batch(qs)
        D.sync_zero_plaintexts(srv, dist, world, comm, torch, "cpu")
        rp = D.RowsReplicatedPipeline(srv, batch, rank, world, dist, torch, "cpu", comm=D.Comm(dist, world))
        rp.submit()
        rp.submit()
        rp.close()
        for t in range(2):
            mine = rp.replies(t).numpy().view(np.uint64)
            for i in range(lo, hi):
                ok &= bool(np.array_equal(mine[i - lo], want(qs[i])))
    elif mode == "hybrid":
        # 2 replica groups x 4 row shards: every group of 4 holds the whole database sharded 4 ways and serves half of the
        # step's queries with the pipelined step inside its own process group (bench.py's hybrid_rows_reference)
        gi, gr, S, groups = D.hybrid_layout(rank, world, 2)
        assert S == 4 and len(groups) == 2 and groups[gi][gr] == rank
        pgs = [dist.new_group(g, backend="gloo") for g in groups]     # every rank creates every group, same order
        srv = OracleShardServer(s, gr, S)
        steps = 2
        idx_all, q_all = [], []
        for t in range(steps):
            a, b = queries_for(t)
            idx_all += a
            q_all += b
        srv.stage_batch(q_all)
        hcomm = D.Comm(dist, S, group=pgs[gi])
        D.sync_zero_plaintexts(srv, dist, S, hcomm, torch, "cpu")
        assert D.packed_exchange_supported(srv, dist, S, hcomm, torch, "cpu")
        bpg = batch // 2
        hp = D.RowsPipeline(srv, bpg, gr, S, dist, torch, "cpu", comm=D.Comm(dist, S, group=pgs[gi]))
        for t in range(steps):
            hp.submit(first=t * batch + gi * bpg)
        hp.flush()
        glo, ghi = D.owned_queries(bpg, gr, S)
        for t in range(steps):
            # only the last two steps' buffers are alive: steps == 2
            mine = hp.replies(t).numpy().view(np.uint64)
            for i in range(glo, ghi):
                ok &= bool(np.array_equal(mine[i - glo], want(q_all[t * batch + gi * bpg + i])))
    elif mode == "slots":
        # slot-sharded step: every rank holds 1 / 8 of the NTT slots of EVERY plaintext (all rows), scans them for all
        # queries of the step and returns the row sums to the query's owner; two pipelined steps + the synchronous form
        srv = OracleShardServer(s, rank, world)
        steps = 2
        idx_all, q_all = [], []
        for t in range(steps):
            a, b = queries_for(t)
            idx_all += a
            q_all += b
        srv.stage_batch(q_all)
        assert D.slots_exchange_supported(srv)
        sb = D.SlotsBuffers(srv, batch, rank, world, torch, "cpu")
        assert sb.per == per and sb.mine == s.orc.k * 4096 // world
        D.run_batch_slots(srv, sb, dist, rank, world, comm)
        mine = sb.replies.numpy().view(np.uint64)
        for i in range(lo, hi):
            ok &= bool(np.array_equal(mine[i - lo], want(q_all[i])))
            ok &= s.client.process_response(p, idx_all[i], mine[i - lo]) == s.item(idx_all[i])
        sp = D.SlotsPipeline(srv, batch, rank, world, dist, torch, "cpu", comm=D.Comm(dist, world))
        for t in range(steps):
            sp.submit(first=t * batch)
        sp.flush()
        for t in range(steps):
            got = sp.replies(t).numpy().view(np.uint64)
            for i in range(lo, hi):
                ok &= bool(np.array_equal(got[i - lo], want(q_all[t * batch + i])))
    else:
        raise ValueError(mode)
    return bool(ok)


def _worker(rank, world, port, rows, per, mode, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out_q.put((rank, world8_check(rank, world, rows, per, mode)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("rows,per,mode", [
    (21, 8, "packed"),        # cfg-3-like: uneven 3/2-row shards, n0 % 8 != 0, ONE full group of 8 per rank
    (13, 3, "packed"),        # shards of 2 and 1 rows, a partial group per rank
    (21, 9, "pipeline"),      # a full group + a partial one per rank, three pipelined steps over both buffer sets
    (21, 2, "replicated"),    # replicated expansion: reduce-scatter only
    (21, 4, "hybrid"),        # 2 replica groups x 4 row shards (6 / 5-row shards inside a group)
    (13, 3, "slots"),         # slot shards: 1 / 8 of the NTT slots of every plaintext per rank, no reduce
], ids=["packed-21rows-per8", "packed-13rows-per3", "pipeline-21rows-per9", "replicated-21rows-per2", "hybrid-2x4-21rows",
        "slots-13rows-per3"])
def test_world_size_eight_over_gloo(rows, per, mode):
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, rows, per, mode, out_q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
        assert p.exitcode == 0
    got = dict(out_q.get(timeout=5) for _ in range(WORLD))
    assert got == {r: True for r in range(WORLD)}
