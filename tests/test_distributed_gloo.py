"""World-size-2 gloo test of the multi-GPU glue on CPU: shard ranges, and that the integer
all-reduce of per-shard partial replies followed by `x mod q_j` equals the full reply.
Per-shard partial replies are produced by the CPU oracle run on the shard's rows only (the recursion of
database.cpp:170-258 over dimension 0 is a sum over its indices)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pir_amd.distributed import shard_range


def test_shard_range_partitions_rows():
    for n in (1, 7, 162, 428, 1639):
        for world in (1, 2, 3, 4, 8):
            cuts = [shard_range(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            for a, b in zip(cuts[:-1], cuts[1:]):
                assert a[1] == b[0]
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, d, dbsize, elem, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from pir_fixtures import PirSetup
        s = PirSetup(dbsize, elem, d, N=4096, plain_bits=24)      # same seeds on every rank
        p = s.params
        q = s.client.create_query_for(p, dbsize - 2)
        # this rank's shard: the plaintexts under its top-level range, multiplied with the matching selectors
        from pir_fixtures import oracle_partial_reply
        lo, hi = shard_range(p.dimensions[0], rank, world)
        stride = 1
        for x in p.dimensions[1:]:
            stride *= x
        rc, sv = s.orc.oblivious_expansion_multi(q, p.dim_sum, s.galois_keys)
        assert rc == 0
        rc, part = oracle_partial_reply(s.orc, s.db_ntt[lo * stride:min(hi * stride, p.num_pt)], p.dimensions, lo, hi, sv)
        assert rc == 0
        t = torch.from_numpy(part.view(np.int64).copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)                 # what RCCL does on the GPUs
        summed = t.numpy().view(np.uint64).copy()
        for j, qj in enumerate(s.orc.moduli[: s.orc.k]):         # the mod-q fix-up kernel's job
            summed[:, :, j, :] %= np.uint64(qj)
        if rank == 0:
            rc, full = s.orc.process_query(s.db_ntt, p.dimensions, q, s.galois_keys)
            ok = bool(np.array_equal(summed, full)) and \
                s.client.process_response(p, dbsize - 2, summed) == s.item(dbsize - 2)
            out_q.put(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("d,dbsize,elem", [(2, 300, 288), (1, 120, 288)])
def test_partial_reply_all_reduce_gloo(d, dbsize, elem):
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, d, dbsize, elem, out_q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert out_q.get(timeout=5) is True
