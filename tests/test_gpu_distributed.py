"""The row-sharded multi-GPU step with the REAL server (libpirgpu) under torch.distributed, world size 2.

The test box has one GPU: both ranks create their shard context on cuda:0 and the collectives run over gloo through
host memory (pir_amd.distributed.Comm stages device tensors when the backend is not RCCL) -- every line of the
step except the RCCL calls themselves is the code `bench.py --gpus N` runs.  Packed exchange (all-gather of packed
column selectors, all-to-all of row selectors, reduce-scatter) and whole-selection-vector exchange, both against
the oracle's full-database replies, plus the world-size-1 degenerate case in-process."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _step(rank, world, dist, items, batch, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import pir_amd
    from pir_amd import distributed as D
    from gpu_helpers import to_product_params
    from pir_fixtures import PirSetup
    s = PirSetup(items, 288, 2, N=4096, plain_bits=24)
    p = s.params
    pp = to_product_params(p)
    shard = D.shard_range(p.dimensions[0], rank, world) if world > 1 else None
    db = pir_amd.PIRDatabase.Create(pp, s.raw, shard=shard)
    srv = pir_amd.PIRServer(db, pp)
    srv.set_galois_keys(s.galois_keys)
    srv.set_concurrency(8)
    indexes = [(items - 1 - 131 * i) % items for i in range(batch)]
    queries = np.stack([s.client.create_query_for(p, i) for i in indexes])
    srv.stage_batch(queries)
    comm = D.Comm(dist, world)
    dev = "cuda:0"
    full = [s.orc.process_query(s.db_ntt, p.dimensions, q, s.galois_keys)[1] for q in queries]
    ok = True
    assert D.packed_exchange_supported(srv, dist, world, comm, torch, dev), srv.scan_info()
    bufs = D.PackedBuffers(srv, batch, rank, world, torch, dev)
    for _ in range(2):                                            # twice: buffers and workers are reused
        D.run_batch_rows_packed(srv, bufs, dist, rank, world, comm)
        lo, hi = D.owned_queries(batch, rank, world)
        mine = bufs.replies.cpu().numpy().view(np.uint64)
        for i in range(lo, hi):
            same = bool(np.array_equal(mine[i - lo], full[i]))
            if not same:
                print("rank %d: packed step, query %d differs" % (rank, i), flush=True)
            ok &= same
    # the PIPELINED step: four consecutive steps over different queries (step t serves the staged queries
    # [t * batch, (t + 1) * batch)), no host synchronisation between the library's streams and the communication
    # stream -- ordering is by events only; replies of step t are read after step t + 2 was submitted or after flush
    steps = 4
    idx_all = indexes + [(items - 7 - 59 * i) % items for i in range(batch * (steps - 1))]
    q_all = np.concatenate([queries, np.stack([s.client.create_query_for(p, i) for i in idx_all[batch:]])])
    srv.stage_batch(q_all)
    pipe = D.RowsPipeline(srv, batch, rank, world, dist, torch, dev)
    lo, hi = D.owned_queries(batch, rank, world)
    seen = {}
    for t in range(steps):
        pipe.submit(first=t * batch)
        if t >= 2:      # the reduce of step t - 2 was queued by submit t - 1: wait for the comm stream, then read
            pipe.streams.side.synchronize()
            seen[t - 2] = pipe.replies(t - 2).cpu().numpy().view(np.uint64).copy()
    pipe.flush()
    seen[steps - 1] = pipe.replies(steps - 1).cpu().numpy().view(np.uint64).copy()
    # step steps - 2 shares its buffer set with step steps - 4 ... its replies were overwritten only by step `steps`,
    # which does not exist: still there
    seen[steps - 2] = pipe.replies(steps - 2).cpu().numpy().view(np.uint64).copy()
    for t in range(steps):
        for i in range(lo, hi):
            g = t * batch + i
            want = full[i] if t == 0 else s.orc.process_query(s.db_ntt, p.dimensions, q_all[g], s.galois_keys)[1]
            same = bool(np.array_equal(seen[t][i - lo], want))
            if not same:
                print("rank %d: pipelined step %d, query %d differs" % (rank, t, i), flush=True)
            ok &= same
    # replicated expansion (bench.py's form of the rows step at two GPUs): every rank expands every query itself on its
    # shard context, the only collective is the reduce-scatter of the partial replies; three steps, two buffer sets
    srv.stage_batch(queries)
    rp = D.RowsReplicatedPipeline(srv, batch, rank, world, dist, torch, dev)
    for _ in range(3):
        rp.submit()
    rp.close()
    for t in (1, 2):
        mine_r = rp.replies(t).cpu().numpy().view(np.uint64)
        for i in range(lo, hi):
            same = bool(np.array_equal(mine_r[i - lo], full[i]))
            if not same:
                print("rank %d: replicated-expansion step %d, query %d differs" % (rank, t, i), flush=True)
            ok &= same
    # hybrid layout (bench.py's `hybrid_rows_reference`), degenerate at two ranks: 2 replica groups of ONE shard each --
    # every rank holds the whole database and serves its half of the queries inside its own one-rank process group
    if world == 2 and batch % 2 == 0:
        gi, gr, S, groups = D.hybrid_layout(rank, world, 2)
        pgs = [dist.new_group(g, backend="gloo") for g in groups]
        dbw = pir_amd.PIRDatabase.Create(pp, s.raw)
        whole = pir_amd.PIRServer.Create(dbw, pp)
        whole.set_galois_keys(s.galois_keys)
        whole.set_concurrency(8)
        whole.stage_batch(q_all)
        bpg = batch // 2
        hp = D.RowsPipeline(whole, bpg, gr, S, dist, torch, dev, comm=D.Comm(dist, S, host_sync=False, group=pgs[gi]))
        hp.submit(first=gi * bpg)
        hp.submit(first=batch + gi * bpg)
        hp.flush()
        for t, base in ((0, 0), (1, batch)):
            mine_h = hp.replies(t).cpu().numpy().view(np.uint64)
            for i in range(bpg):
                g = base + gi * bpg + i
                want = full[g] if g < batch else s.orc.process_query(s.db_ntt, p.dimensions, q_all[g], s.galois_keys)[1]
                same = bool(np.array_equal(mine_h[i], want))
                if not same:
                    print("rank %d: hybrid step %d, query %d differs" % (rank, t, g), flush=True)
                ok &= same
        dbw.close()
    srv.stage_batch(queries)
    sv_all = torch.empty((batch, p.dim_sum, 2, srv.k, srv.N), dtype=torch.int64, device=dev)
    replies = torch.empty((batch, db.reply_ct_count(), 2, srv.k, srv.N), dtype=torch.int64, device=dev)
    D.run_batch_query_parallel(srv, sv_all, replies, dist, rank, world, comm)
    allr = replies.cpu().numpy().view(np.uint64)
    for i in range(batch):
        same = bool(np.array_equal(allr[i], full[i]))
        if not same:
            print("rank %d: selection-vector step, query %d differs" % (rank, i), flush=True)
        ok &= same
    out.append(ok)
    db.close()


def _worker(rank, world, port, items, batch, out_q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = []
        _step(rank, world, dist, items, batch, out)
        out_q.put((rank, out[0]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("items,batch", [(10800, 4), (10800, 18)])   # 270 plaintexts = 17 x 16: shards of 8 and 9 rows
def test_row_sharded_step_two_ranks_one_gpu(items, batch):
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, items, batch, out_q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    assert dict(out_q.get(timeout=5) for _ in range(2)) == {0: True, 1: True}


def test_row_sharded_step_world_size_one():
    out = []
    _step(0, 1, None, 10800, 3, out)
    assert out == [True]


def test_batch_replies_into_a_caller_buffer():
    """pirgpu_batch_set_reply_buffer: groups write their replies straight into the caller's device buffer (the send
    buffer of the multi-GPU reduce); a buffer too small for the batch is refused before anything runs."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import pir_amd
    from pir_amd.server import PirGpuError
    from gpu_helpers import to_product_params
    from pir_fixtures import PirSetup
    s = PirSetup(10800, 288, 2, N=4096, plain_bits=24)
    pp = to_product_params(s.params)
    db = pir_amd.PIRDatabase.Create(pp, s.raw)
    srv = pir_amd.PIRServer(db, pp)
    srv.set_galois_keys(s.galois_keys)
    srv.set_concurrency(8)
    batch = 11                                             # one full group of 8 and a ragged one
    queries = [s.client.create_query_for(s.params, (7 + 977 * i) % 10800) for i in range(batch)]
    srv.stage_batch(queries)
    srv.run_batch()
    plain = srv.fetch_batch()
    n = db.reply_ct_count()
    mine = torch.zeros((batch, n, 2, srv.k, srv.N), dtype=torch.int64, device="cuda:0")
    srv.batch_set_reply_buffer(mine.data_ptr(), batch * n)
    srv.stage_batch(queries)
    srv.run_batch()
    srv.sync()
    assert np.array_equal(mine.cpu().numpy().view(np.uint64), plain)
    assert np.array_equal(srv.fetch_batch(), plain)        # fetch reads where the batch wrote
    srv.batch_set_reply_buffer(mine.data_ptr(), batch * n - 1)
    srv.stage_batch(queries)
    with pytest.raises(PirGpuError) as e:
        srv.run_batch()
    assert e.value.code == 3 and "reply buffer" in e.value.message
    srv.batch_set_reply_buffer(0, 0)                       # back to the context's own buffer
    srv.stage_batch(queries)
    srv.run_batch()
    assert np.array_equal(srv.fetch_batch(), plain)
    db.close()


def test_batch_replies_downloaded_group_by_group():
    """pirgpu_batch_set_host_replies: every group of a batch sends its replies to the pinned host buffer on its own
    stream as soon as they exist; pirgpu_batch_fetch into that buffer only waits.  Same replies as the plain fetch,
    for a batch of two full groups and a ragged one; a batch larger than the buffer falls back to the plain download."""
    import ctypes as C
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import pir_amd
    from gpu_helpers import to_product_params
    from pir_fixtures import PirSetup
    s = PirSetup(10800, 288, 2, N=4096, plain_bits=24)
    pp = to_product_params(s.params)
    db = pir_amd.PIRDatabase.Create(pp, s.raw)
    srv = pir_amd.PIRServer(db, pp)
    srv.set_galois_keys(s.galois_keys)
    srv.set_concurrency(16)
    batch = 19
    queries = [s.client.create_query_for(s.params, (11 + 523 * i) % 10800) for i in range(batch)]
    srv.stage_batch(queries)
    srv.run_batch()
    plain = srv.fetch_batch()
    n, words = db.reply_ct_count(), 2 * srv.k * srv.N
    lib, h = srv.lib, db.handle
    host = lib.pirgpu_host_reply_buffer(h, batch)
    assert host
    view = np.ctypeslib.as_array(C.cast(host, C.POINTER(C.c_uint64)), shape=(batch, n, 2, srv.k, srv.N))
    for cap in (batch * n, batch * n - 1):          # fits / does not fit: group-wise download / plain download
        view[:] = 0
        assert lib.pirgpu_batch_set_host_replies(h, C.c_void_p(host), cap) == 0
        srv.stage_batch(queries)
        srv.run_batch()
        cnt = C.c_uint64(0)
        assert lib.pirgpu_batch_fetch(h, C.cast(host, C.POINTER(C.c_uint64)), batch * n, C.byref(cnt)) == 0
        assert cnt.value == batch * n
        assert np.array_equal(view, plain), cap
    # pirgpu_batch_next_host_replies: the groups are reported in order as their replies land; what has been reported is
    # complete in the host buffer before the batch as a whole is
    view[:] = 0
    assert lib.pirgpu_batch_set_host_replies(h, C.c_void_p(host), batch * n) == 0
    srv.stage_batch(queries)
    srv.run_batch()
    seen, ready = [], C.c_uint32(0)
    while not seen or seen[-1] < batch:
        assert lib.pirgpu_batch_next_host_replies(h, C.byref(ready)) == 0
        assert ready.value > (seen[-1] if seen else 0)
        assert np.array_equal(view[:ready.value], plain[:ready.value])
        seen.append(ready.value)
    assert seen == [8, 16, 19]
    assert lib.pirgpu_batch_next_host_replies(h, C.byref(ready)) == 0 and ready.value == batch   # nothing left: the total
    cnt = C.c_uint64(0)
    assert lib.pirgpu_batch_fetch(h, C.cast(host, C.POINTER(C.c_uint64)), batch * n, C.byref(cnt)) == 0
    assert lib.pirgpu_batch_set_host_replies(h, None, 0) == 0
    srv.stage_batch(queries)
    srv.run_batch()
    assert np.array_equal(srv.fetch_batch(), plain)
    assert words == view.shape[2] * view.shape[3] * view.shape[4]
    db.close()


def test_bench_eight_ranks_sharing_one_gpu():
    """`bench.py --gpus 8` end to end at world size 8 on ONE GPU (PIRGPU_BENCH_SHARE_GPU=1: all ranks use device 0, the
    collectives go over gloo): the launcher, eight row shards of 10 / 11 rows of an 81 x 81 matrix (uneven, n0 % 8 != 0),
    one full group of 8 queries per rank, both forms of the row-sharded step timed and the faster one reported, the
    replica and hybrid 2 x 4 reference legs.  Not a multi-GPU measurement (the line says so) -- a rehearsal of every
    line of the driver's 8-GPU run except the RCCL calls themselves."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["PIRGPU_BENCH_SHARE_GPU"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--log-items", "18", "--steps", "2",
                        "--warmup", "1", "--latency-runs", "2", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1                                   # exactly rank 0's line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["rccl_ranks"] == 0 and "gloo" in j["backend"]
    assert j["config"]["queries_per_step"] == 64 and j["config"]["queries_per_step_per_gpu"] == 8
    assert j["value"] > 0 and j["scaling"] == "strong"
    tune = j["exchange_autotune"]
    # all three forms of the sharded step ran: slot shards FIRST (the candidate whose budget meets north_star), then rows +
    # replicated expansion, then rows + packed exchange -- each with its wall seconds in the line
    assert tune["order"] == ["slots", "replicated", "packed"] and list(tune["ms_per_step"]) == tune["order"]
    assert set(tune["wall_s"]) == set(tune["order"]) and all(v > 0 for v in tune["wall_s"].values())
    assert set(tune["ms_per_step"]) == {"replicated", "packed", "slots"} and tune["chosen"] == j["config"]["exchange"]
    assert all(v and v > 0 for v in tune["ms_per_step"].values()), tune
    assert j["replicas_reference"]["value"] > 0
    assert j["hybrid_rows_reference"]["value"] > 0 and j["hybrid_rows_reference"]["row_shards_per_group"] == 4
    assert "extras_aborted" not in j


def test_bench_a_stalled_candidate_leaves_the_slots_headline():
    """`bench.py --gpus 2` with the LAST candidate form made to stall (PIRGPU_TEST_STALL=packed: an hour of sleep in every
    measured step -- what a collective that never completes looks like to the job): the slot-sharded step was measured
    first and stays the headline, `replicated` ran inside its own slice of PIRGPU_EXTRAS_TIMEOUT_S, the stalled form is cut
    off at the end of ITS slice, rank 0 prints the line (naming the stage) and every rank exits with the watchdog's code 3.
    Fresh child processes only (pir_amd/launcher.py): nothing re-executes a process that has touched the GPU."""
    import json
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PIRGPU_BENCH_SHARE_GPU="1", PIRGPU_TEST_STALL="packed", PIRGPU_EXTRAS_TIMEOUT_S="120")
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--log-items", "16", "--steps", "2",
                        "--warmup", "1", "--latency-runs", "2", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=900)
    wall = time.monotonic() - t0
    assert p.returncode == 3, (p.returncode, p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    tune = j["exchange_autotune"]
    assert tune["order"] == ["slots", "replicated", "packed"] and tune["watchdog_slice_s"] == 40.0
    assert tune["ms_per_step"]["slots"] > 0 and tune["ms_per_step"]["replicated"] > 0 and "packed" not in tune["ms_per_step"]
    assert j["config"]["exchange"] == tune["chosen"] and tune["chosen"] in ("slots", "replicated")
    assert j["value"] > 0 and "candidate packed" in j["extras_aborted"]
    assert wall < 600          # the stalled form cost its own slice (40 s), not an hour
