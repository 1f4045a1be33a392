"""Multi-GPU glue (one process per GPU, torch.distributed over RCCL).  Two ways of sharding the database, see DESIGN.md
section 7 (neither is in the reference, which is single-threaded and single-process):

  * SLOT shards (run_batch_slots, SlotsPipeline): every rank holds 1 / G of the NTT slots of every plaintext; per step an
    all-to-all hands it its slots of every query's packed column selectors, it scans all rows for all queries, a second
    all-to-all returns the row sums to the rank that expanded the query, which finishes its own queries -- no reduce;
  * ROW shards (run_batch_rows_packed, RowsPipeline, RowsReplicatedPipeline, run_batch_query_parallel): every rank holds
    1 / G of the rows; the exchange of what every shard needs from a query, and the sum of the per-shard partial replies.

The server object is duck-typed (pir_amd.PIRServer; the CPU tests drive the same code with an oracle-backed
stand-in), the collectives go through `Comm`, which runs them on the tensors' own device (RCCL) or -- for the
world-size-2 `gloo` tests -- stages them through host memory.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

GROUP = 8          # queries per packed group = queries sharing one database pass (kMaxMfmaQueries)


def shard_range(n_top: int, rank: int, world: int) -> Tuple[int, int]:
    """Top-level index range [begin, end) of dimension 0 held by `rank` (contiguous, balanced)."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    return (n_top * rank) // world, (n_top * (rank + 1)) // world


def row_cuts(n_top: int, world: int) -> List[int]:
    """shard_range of every rank as one non-decreasing list of world + 1 cuts."""
    return [(n_top * r) // world for r in range(world + 1)]


def check_sum_fits(q_max: int, world: int) -> None:
    """Partial replies are canonical residues < q_j; their integer sum over the ranks must not wrap 64 bits
    (int64 tensors wrap like uint64, so the bound is 2^64)."""
    if world * (q_max - 1) >= 1 << 64:
        raise ValueError("world size %d too large for %d-bit moduli: the 64-bit sum of partial replies would wrap"
                         % (world, q_max.bit_length()))


def hybrid_layout(rank: int, world: int, replica_groups: int):
    """R replica groups of S = world / R row shards each (ranks g*S .. g*S+S-1 form group g): returns
    (group index, rank inside the group, S, [ranks of every group])."""
    if replica_groups < 1 or world % replica_groups:
        raise ValueError("world size must be a multiple of the number of replica groups")
    S = world // replica_groups
    groups = [list(range(g * S, (g + 1) * S)) for g in range(replica_groups)]
    return rank // S, rank % S, S, groups


def owned_queries(count: int, rank: int, world: int):
    """Queries of a batch this rank expands: the contiguous block [begin, end) (count % world == 0)."""
    if count % world:
        raise ValueError("batch size must be a multiple of the world size for query-parallel expansion")
    per = count // world
    return rank * per, (rank + 1) * per


class Comm:
    """The four collectives of a step.  backend 'nccl' (RCCL): on the tensors' device, on torch's current stream.
    With host_sync (the synchronous steps: run_batch_rows_packed, run_batch_query_parallel) every collective is followed
    by a stream synchronise, because the library runs on its own HIP streams; RowsPipeline passes host_sync=False and
    orders streams with events instead.  Any other backend (gloo in the tests): tensors are staged through host
    memory, so the same glue runs with two processes on one GPU or on CPU."""

    # Largest per-peer message handed to one RCCL call.  RCCL 2.26.6 (ROCm 7.0) copies only the first half of an
    # all_to_all_single message above 1 GiB (tools/experiments/r03_a2a_probe.py, MI355X: every size <= 1024 MiB exact, 1100 MiB
    # and up wrong from the middle on); larger messages are cut into pieces at sub-block boundaries.  Real 8-GPU
    # steps stay far below (21 MB per peer at cfg 3); a forced single-rank run of the same step does not.
    MAX_MESSAGE_BYTES = 512 << 20
    HARD_MESSAGE_BYTES = 1 << 30     # above this a single RCCL message is not trusted at all (all_to_all raises)

    def __init__(self, dist, world: int, host_sync: bool = True, group=None):
        """world: ranks taking part (the size of `group`, a torch.distributed process group; None = the default one)."""
        self.dist, self.world = dist, world
        self.group = group
        self.host_sync = host_sync
        import os
        if os.environ.get("PIRGPU_MAX_COLLECTIVE_MB"):      # tests: force the piecewise paths with small messages
            self.MAX_MESSAGE_BYTES = max(1, int(os.environ["PIRGPU_MAX_COLLECTIVE_MB"])) << 20
        # RCCL process group (also a forced single-rank one): collectives run on the device
        self.device_native = dist is not None and dist.is_initialized() and dist.get_backend() == "nccl"

    def _sync(self, t):
        if t.is_cuda and self.host_sync:
            import torch
            torch.cuda.current_stream(t.device).synchronize()

    def wait(self, t):
        """Host wait for the collectives queued so far on torch's current stream of t's device (whatever host_sync)."""
        if t.is_cuda:
            import torch
            torch.cuda.current_stream(t.device).synchronize()

    def all_gather_inplace(self, full, rank: int):
        """full: [world, ...]; every rank has filled full[rank]."""
        if self.world == 1 and not self.device_native:
            return
        d = self.dist
        if self.device_native:
            flat = full.view(self.world, -1)
            n, limit = flat.shape[1], max(1, self.MAX_MESSAGE_BYTES // full.element_size())
            if n <= limit:
                d.all_gather_into_tensor(full.view(-1), full[rank].reshape(-1), group=self.group)
            else:
                for o in range(0, n, limit):
                    e = min(n, o + limit)
                    d.all_gather([flat[r, o:e] for r in range(self.world)], flat[rank, o:e], group=self.group)
            self._sync(full)
            return
        mine = full[rank].cpu().contiguous()
        parts = [mine.new_empty(mine.shape) for _ in range(self.world)]
        d.all_gather(parts, mine, group=self.group)
        for r, p in enumerate(parts):
            full[r].copy_(p)
        self._sync(full)

    def all_to_all(self, recv, send, recv_splits, send_splits, units: int = 1):
        """1-D tensors; split sizes in elements.  units: every per-peer block consists of `units` equal sub-blocks
        (the queries of a step); a block larger than MAX_MESSAGE_BYTES is exchanged in pieces of whole sub-blocks."""
        if self.world == 1 and not self.device_native:
            recv.copy_(send)
            self._sync(recv)
            return
        d = self.dist
        if self.device_native:
            limit = max(1, self.MAX_MESSAGE_BYTES // recv.element_size())
            biggest = max(max(recv_splits), max(send_splits))
            need = -(-biggest // limit)                      # pieces a per-peer block has to be cut into
            pieces = min(max(1, units), need)
            if pieces > 1 and any(x % units for x in list(recv_splits) + list(send_splits)):
                raise ValueError("all_to_all: split sizes must be multiples of `units` to be exchanged in pieces")
            # a block can only be cut at sub-block boundaries: refuse a message that would still exceed what RCCL
            # moves correctly (2.26.6 silently corrupts single messages above 1 GiB) instead of sending it
            hard = max(1, self.HARD_MESSAGE_BYTES // recv.element_size())
            if -(-biggest // max(1, units)) * -(-max(1, units) // pieces) > hard:
                raise ValueError("all_to_all: a per-peer block of %d elements cannot be cut into pieces of at most %d "
                                 "elements at its %d sub-block boundaries (HARD_MESSAGE_BYTES)" % (biggest, hard, max(1, units)))
            if pieces <= 1:
                d.all_to_all_single(recv, send, list(recv_splits), list(send_splits), group=self.group)
            else:
                ro, so = [0], [0]
                for x in recv_splits:
                    ro.append(ro[-1] + x)
                for x in send_splits:
                    so.append(so[-1] + x)
                for pc in range(pieces):
                    u0, u1 = units * pc // pieces, units * (pc + 1) // pieces
                    outs = [recv[ro[r] + recv_splits[r] // units * u0: ro[r] + recv_splits[r] // units * u1]
                            for r in range(self.world)]
                    ins = [send[so[r] + send_splits[r] // units * u0: so[r] + send_splits[r] // units * u1]
                           for r in range(self.world)]
                    d.all_to_all(outs, ins, group=self.group)
            self._sync(recv)
            return
        s, r = send.cpu(), recv.new_empty(recv.shape, device="cpu")
        d.all_to_all_single(r, s, list(recv_splits), list(send_splits), group=self.group)
        recv.copy_(r)
        self._sync(recv)

    def reduce_scatter_sum(self, out, full, rank: int):
        """full: [world * n] int64 partial sums, out: [n] = sum over ranks of full[rank*n:(rank+1)*n]."""
        if self.world == 1 and not self.device_native:
            out.copy_(full.view(-1)[: out.numel()].view(out.shape))
            self._sync(out)
            return
        d = self.dist
        if self.device_native:
            o1, f2 = out.view(-1), full.view(self.world, -1)
            n, limit = o1.numel(), max(1, self.MAX_MESSAGE_BYTES // out.element_size())
            if n <= limit:
                d.reduce_scatter_tensor(o1, full.view(-1), op=d.ReduceOp.SUM, group=self.group)
            else:
                for o in range(0, n, limit):
                    e = min(n, o + limit)
                    d.reduce_scatter(o1[o:e], [f2[r, o:e] for r in range(self.world)], op=d.ReduceOp.SUM,
                                     group=self.group)
            self._sync(out)
            return
        c = full.cpu()
        d.all_reduce(c, op=d.ReduceOp.SUM, group=self.group)   # gloo has no reduce-scatter: reduce everything, keep the slice
        n = out.numel()
        out.copy_(c.view(-1)[rank * n:(rank + 1) * n].view(out.shape))
        self._sync(out)

    def all_reduce_sum(self, t):
        if self.world == 1 and not self.device_native:
            return
        d = self.dist
        if self.device_native:
            d.all_reduce(t, op=d.ReduceOp.SUM, group=self.group)
            self._sync(t)
            return
        c = t.cpu()
        d.all_reduce(c, op=d.ReduceOp.SUM, group=self.group)
        t.copy_(c)
        self._sync(t)


def all_reduce_reply(server, reply_tensor, dist, comm: Optional[Comm] = None) -> None:
    """Sum the ranks' partial replies of the last single query in place and reduce mod q_j.
    reply_tensor: int64 tensor [reply_cts, 2, k, N] owned by the caller (on the server's device)."""
    comm = comm or Comm(dist, dist.get_world_size())
    server.reply_copy_to_device(reply_tensor.data_ptr())     # waits for this rank's kernels
    comm.all_reduce_sum(reply_tensor)                         # RCCL over xGMI
    server.reduce_fixup_device(reply_tensor.data_ptr())       # x mod q_j on the GPU


def all_reduce_batch_replies(server, reply_tensor, dist, comm: Optional[Comm] = None) -> None:
    """Batch-mode counterpart of all_reduce_reply: reply_tensor int64 [count, reply_cts, 2, k, N]."""
    comm = comm or Comm(dist, dist.get_world_size())
    server.batch_reply_copy_to_device(reply_tensor.data_ptr())   # waits for every worker stream
    comm.all_reduce_sum(reply_tensor)
    server.reduce_fixup_device_n(reply_tensor.data_ptr(), reply_tensor.shape[0] * reply_tensor.shape[1])


def sync_zero_plaintexts(server, dist, world: int, comm: Optional[Comm] = None, torch=None, device=None) -> int:
    """Makes the reference's transparent-ciphertext failure a COLLECTIVE decision (once, after the database is loaded):
    the reference fails every query as soon as any plaintext of the whole database is identically zero (SEAL's
    logic_error through database.cpp:313-315), but a row shard only sees its own rows -- without this, the one rank that
    holds the zero plaintext would raise in the middle of a step and leave its peers blocked in the next collective.
    One all-reduce of the shards' counts; every rank tells its context how many the others hold.  Returns the total."""
    mine = int(server.zero_plaintexts())
    total = mine
    if world > 1:
        t = torch.tensor([mine], dtype=torch.int64, device=device)
        (comm or Comm(dist, world)).all_reduce_sum(t)
        total = int(t.item())
    server.set_remote_zero_plaintexts(total - mine)
    return total


def run_batch_query_parallel(server, sv_all, replies, dist, rank: int, world: int, comm: Optional[Comm] = None) -> None:
    """One step over a staged batch on `world` GPUs holding row shards, exchanging whole u64 selection vectors
    (any d, any scan kernel; 2 k N 8 dim_sum bytes per query to every rank):

      1. every rank expands only its own block of the batch's queries straight into its slice of `sv_all`;
      2. one all-gather makes every query's NTT-form selection vector available everywhere;
      3. every rank multiplies all queries against its row shard (shared database passes);
      4. the partial replies are summed with one all-reduce and reduced mod q_j.

    sv_all:  int64 tensor [count, dim_sum, 2, k, N]   (all-gather buffer)
    replies: int64 tensor [count, reply_cts, 2, k, N] (all-reduce buffer; every rank ends with every reply)
    """
    comm = comm or Comm(dist, world)
    server.check_ready()        # identical on every rank after sync_zero_plaintexts: nobody enters a collective alone
    count = sv_all.shape[0]
    lo, hi = owned_queries(count, rank, world)
    server.batch_expand(lo, hi - lo, sv_all[lo].data_ptr())          # synchronous
    comm.all_gather_inplace(sv_all.view(world, -1), rank)
    server.batch_run_selectors(sv_all.data_ptr(), count)
    all_reduce_batch_replies(server, replies, dist, comm)


class PackedBuffers:
    """Device buffers of run_batch_rows_packed for one (server, batch size, world)."""

    def __init__(self, server, batch: int, rank: int, world: int, torch, device):
        if batch % world:
            raise ValueError("batch size must be a multiple of the world size")
        self.per = batch // world
        self.groups = (self.per + GROUP - 1) // GROUP
        self.cuts = row_cuts(server.params.dimensions[0], world)
        self.sel_bytes = server.packed_selector_bytes()
        k, N = server.k, server.N
        self.ctw = 2 * k * N
        self.my_rows = self.cuts[rank + 1] - self.cuts[rank]
        n0 = server.params.dimensions[0]
        reply_cts = server.db.reply_ct_count()
        self.packed = torch.empty((world, self.groups, max(self.sel_bytes, 1)), dtype=torch.uint8, device=device)
        self.rows_send = torch.empty((self.per * n0 * self.ctw,), dtype=torch.int64, device=device)
        self.rows_recv = torch.empty((batch * self.my_rows * self.ctw,), dtype=torch.int64, device=device)
        self.partial = torch.empty((batch, reply_cts, 2, k, N), dtype=torch.int64, device=device)
        self.replies = torch.empty((self.per, reply_cts, 2, k, N), dtype=torch.int64, device=device)
        self.send_splits = [self.per * (self.cuts[s + 1] - self.cuts[s]) * self.ctw for s in range(world)]
        self.recv_splits = [self.per * self.my_rows * self.ctw] * world
        # Row selectors cross the links in 5 bytes per residue where the moduli allow (below 2^40: cfg 2 / 3): every 4
        # words are 5 dwords, so the per-rank pieces of the all-to-all are cut at the same places (x 5 / 4).  They are
        # 15 % of a rank's ingest at 8 GPUs and 45 % at 2; PIRGPU_ROWS_PACK40=0 sends u64.
        self.rows40 = (os.environ.get("PIRGPU_ROWS_PACK40", "1") != "0" and hasattr(server, "pack40_supported")
                       and server.pack40_supported())
        if self.rows40:
            self.rows_send40 = torch.empty((self.rows_send.numel() * 5 // 4,), dtype=torch.int32, device=device)
            self.rows_recv40 = torch.empty((self.rows_recv.numel() * 5 // 4,), dtype=torch.int32, device=device)
            self.send_splits40 = [v * 5 // 4 for v in self.send_splits]
            self.recv_splits40 = [v * 5 // 4 for v in self.recv_splits]

    def exchange_bytes_per_query(self, world: int) -> float:
        """Bytes a rank receives per query of the batch (packed column selectors from the other ranks + its rows)."""
        return (world - 1) / world * (self.sel_bytes * self.groups / self.per
                                      + self.my_rows * self.ctw * (5 if self.rows40 else 8))

    # -- the rows part of the exchange, in whichever form this buffer set uses -------------------------------------
    def pack_rows(self, server, stream: int = 0) -> None:
        """After the expansion wrote rows_send: its 5-byte form (queued on `stream`; 0 = the library's main stream)."""
        if self.rows40:
            server.pack40_async(self.rows_send.data_ptr(), self.rows_send40.data_ptr(), self.rows_send.numel(), stream)

    def exchange_rows(self, comm) -> None:
        if self.rows40:
            comm.all_to_all(self.rows_recv40, self.rows_send40, self.recv_splits40, self.send_splits40, units=self.per)
        else:
            comm.all_to_all(self.rows_recv, self.rows_send, self.recv_splits, self.send_splits, units=self.per)

    def unpack_rows(self, server, stream: int = 0) -> None:
        """Before the multiply reads rows_recv: back from the 5-byte form."""
        if self.rows40:
            server.unpack40_async(self.rows_recv40.data_ptr(), self.rows_recv.data_ptr(), self.rows_recv.numel(), stream)


def packed_exchange_supported(server, dist, world: int, comm: Optional[Comm] = None, torch=None, device=None) -> bool:
    """True when every rank's shard can take the packed exchange (d = 2, int8-MFMA scan active on the shard)."""
    mine = 1 if server.packed_selector_bytes() > 0 else 0
    if world == 1:
        return bool(mine)
    t = torch.tensor([mine], dtype=torch.int64, device=device)
    (comm or Comm(dist, world)).all_reduce_sum(t)
    return int(t.item()) == world


def run_batch_rows_packed(server, bufs: PackedBuffers, dist, rank: int, world: int, comm: Optional[Comm] = None) -> dict:
    """One step over a staged batch on `world` GPUs holding row shards, with the PACKED exchange (d = 2):

      1. every rank expands its own `per` queries (groups of 8 expanded together), packs their column selectors
         into the scan's B-operand layout (L signed bytes per residue) and lays their row selectors out by owner;
      2. all-gather of the packed column selectors; all-to-all of the row selectors (a rank receives only its rows);
      3. every rank scans its row shard once per group of 8 queries and runs the upper level on its rows;
      4. reduce-scatter of the partial replies: rank r ends with the finished replies of the queries it expanded
         (reply i answers query i, reference server.cpp:60-63), then x mod q_j.

    Per query a rank receives (world-1)/world * (14.4 MB + 2.6 MB/world) at cfg 3 instead of 42 MB of u64 selectors.
    """
    comm = comm or Comm(dist, world)
    server.check_ready()        # identical on every rank after sync_zero_plaintexts: nobody enters a collective alone
    import time
    t = [time.perf_counter()]
    lo, hi = owned_queries(bufs.per * world, rank, world)
    server.batch_expand_packed(lo, bufs.per, bufs.packed[rank].data_ptr(), bufs.rows_send.data_ptr(), bufs.cuts)
    bufs.pack_rows(server)
    server.sync()
    t.append(time.perf_counter())
    comm.all_gather_inplace(bufs.packed, rank)
    bufs.exchange_rows(comm)
    t.append(time.perf_counter())
    bufs.unpack_rows(server)
    server.fork()               # the lanes start behind the main stream's unpacking
    server.batch_run_packed(bufs.packed.data_ptr(), world, bufs.per, bufs.rows_recv.data_ptr())
    server.batch_reply_copy_to_device(bufs.partial.data_ptr())
    t.append(time.perf_counter())
    comm.reduce_scatter_sum(bufs.replies, bufs.partial, rank)
    server.reduce_fixup_device_n(bufs.replies.data_ptr(), bufs.replies.shape[0] * bufs.replies.shape[1])
    t.append(time.perf_counter())
    # every phase above ends with a host wait in this (synchronous) form of the step: its serial phase times
    return {"expand_ms": (t[1] - t[0]) * 1e3, "exchange_ms": (t[2] - t[1]) * 1e3, "multiply_ms": (t[3] - t[2]) * 1e3,
            "reduce_ms": (t[4] - t[3]) * 1e3}


class _NoStreams:
    """Stand-in for the stream plumbing when the tensors live on the CPU (oracle-backed server in the gloo tests)."""

    class _Ctx:
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    def comm(self):
        return self._Ctx()

    def comm_after_main(self):
        pass

    def record_exchange(self, b):
        pass

    def main_after_exchange(self, b):
        pass

    def comm_handle(self):
        return 0

    def synchronize(self):
        pass


class _GpuStreams:
    """The library's main stream (wrapped, not owned) and one communication stream of ours; the two are ordered
    against each other with events only -- the host never waits inside a step."""

    def __init__(self, server, torch, device):
        self.torch = torch
        self.main = torch.cuda.ExternalStream(server.stream_handle(), device=device)
        self.side = torch.cuda.Stream(device=device)
        self.ev_x = [torch.cuda.Event(), torch.cuda.Event()]

    def comm(self):
        return self.torch.cuda.stream(self.side)

    def comm_after_main(self):
        self.side.wait_stream(self.main)

    def record_exchange(self, b):
        self.ev_x[b].record(self.side)

    def main_after_exchange(self, b):
        self.main.wait_event(self.ev_x[b])

    def comm_handle(self):
        return int(self.side.cuda_stream)

    def synchronize(self):
        self.side.synchronize()
        self.main.synchronize()


class RowsPipeline:
    """The row-sharded step of run_batch_rows_packed, PIPELINED over consecutive steps and free of host waits:

        lanes (library):  E_s   M_(s-1)   E_(s+1)   M_s      ...      E = expand + pack own queries, M = scans + upper level
        comm stream:         X_s    R_(s-1)    X_(s+1)   R_s ...      X = all-gather + all-to-all, R = reduce-scatter + mod q

    so the exchange of step s (the long pole: every rank receives the packed column selectors of every query) runs under
    the multiply of step s-1 and the expansion of step s+1 instead of between them.  Two buffer sets; the library's
    streams and the communication stream are ordered with events (pirgpu_join / pirgpu_fork on the library side,
    ExternalStream / wait_stream on ours); nothing in submit() blocks the host under RCCL.  Reply i of a step still
    answers query i (reference server.cpp:60-63): rank r ends with the replies of the queries it expanded, in
    `replies(step)`.  submit() only QUEUES work (the reduce of a step runs on the communication stream, which the
    caller's stream does not follow), and the reduce of step s + 2 overwrites the buffer of step s: the replies of a
    step are valid -- and must be consumed -- after `flush()` and before the next-but-one submit; flush() is the only
    safe point.

    submit(first) expands the staged queries [first + rank * per, first + (rank + 1) * per) -- with several batches
    staged back to back, consecutive steps can serve different queries."""

    def __init__(self, server, batch: int, rank: int, world: int, dist, torch, device, comm: Optional[Comm] = None):
        self.server, self.rank, self.world, self.dist = server, rank, world, dist
        self.sets = [PackedBuffers(server, batch, rank, world, torch, device) for _ in range(2)]
        self.per = self.sets[0].per
        on_gpu = str(device).startswith("cuda")
        self.comm = comm or Comm(dist, world, host_sync=False)
        self.comm.host_sync = False
        self.streams = _GpuStreams(server, torch, device) if on_gpu else _NoStreams()
        self.step = 0            # steps submitted
        self.pending = None      # buffer set whose multiply + reduce is still to be queued
        self.n_reply_cts = self.sets[0].replies.shape[0] * self.sets[0].replies.shape[1]

    def _finish(self, b: int) -> None:
        """Multiply + reduce of the step whose exchange went into set b."""
        srv, st, bufs = self.server, self.streams, self.sets[b]
        st.main_after_exchange(b)             # the main stream waits for X of that step ...
        bufs.unpack_rows(srv)                 # (row selectors back from their 5-byte form, on the main stream)
        srv.fork()                            # ... and with it the lanes
        srv.batch_run_packed(bufs.packed.data_ptr(), self.world, bufs.per, bufs.rows_recv.data_ptr())
        srv.batch_reply_copy_to_device_async(bufs.partial.data_ptr())      # join + copy on the main stream
        st.comm_after_main()
        with st.comm():
            self.comm.reduce_scatter_sum(bufs.replies, bufs.partial, self.rank)
            srv.reduce_fixup_device_async(bufs.replies.data_ptr(), self.n_reply_cts, st.comm_handle())

    def submit(self, first: int = 0) -> None:
        srv, st = self.server, self.streams
        srv.check_ready()
        b = self.step & 1
        bufs = self.sets[b]
        # the lanes must be done with the multiply that last read this set (two steps ago) before it is refilled
        srv.fork()
        srv.batch_expand_packed_async(first + self.rank * bufs.per, bufs.per, bufs.packed[self.rank].data_ptr(),
                                      bufs.rows_send.data_ptr(), bufs.cuts)
        srv.join()
        bufs.pack_rows(srv)                   # on the main stream, behind the expansion it has just been joined to
        st.comm_after_main()
        with st.comm():
            self.comm.all_gather_inplace(bufs.packed, self.rank)
            bufs.exchange_rows(self.comm)
            st.record_exchange(b)
        if self.pending is not None:
            self._finish(self.pending)
        self.pending = b
        self.step += 1

    def flush(self) -> None:
        """Queues the multiply + reduce of the last submitted step and waits for everything."""
        if self.pending is not None:
            self._finish(self.pending)
            self.pending = None
        self.streams.synchronize()
        self.server.sync()

    def replies(self, step: int):
        """The replies tensor of `step` (0-based submit index): [per, reply_cts, 2, k, N] int64."""
        return self.sets[step & 1].replies


class RowsReplicatedPipeline:
    """Row-sharded database with the oblivious expansion REPLICATED: every rank expands every query of the step itself
    (the plain batch pipeline on its own shard), so no selectors cross GPUs at all -- the only collective is the
    reduce-scatter (SUM) of the per-shard partial replies + x mod q_j, the literal reading of "the database shards
    row-wise ... per-shard reply ciphertexts summed by an RCCL reduce".  It costs every rank the whole expansion work
    (7.5 of a single GPU's 11.9 ms per step of 64 queries at cfg 3), so it scales only as far as the multiply does --
    but at TWO GPUs, where the packed exchange would push 0.8 GB per step through the one xGMI link between them
    (about 16 ms against 6.9 ms of compute), it is the better form: 7.5 + 3.05 ms of compute, 34 MB to reduce.  The
    groups write their replies straight into the reduce's send buffer (pirgpu_batch_set_reply_buffer); the reduce of
    step s runs on the communication stream under the compute of step s + 1 (two buffer sets, event-ordered
    streams, no host waits); rank r ends with the replies of queries [r * per, (r + 1) * per) in `replies(step)`."""

    def __init__(self, server, batch: int, rank: int, world: int, dist, torch, device, comm: Optional[Comm] = None):
        if batch % world:
            raise ValueError("batch size must be a multiple of the world size")
        self.server, self.rank, self.world, self.batch = server, rank, world, batch
        self.per = batch // world
        k, N = server.k, server.N
        reply_cts = server.db.reply_ct_count()
        self.partial = [torch.empty((batch, reply_cts, 2, k, N), dtype=torch.int64, device=device) for _ in range(2)]
        self._replies = [torch.empty((self.per, reply_cts, 2, k, N), dtype=torch.int64, device=device) for _ in range(2)]
        self.comm = comm or Comm(dist, world, host_sync=False)
        self.comm.host_sync = False
        on_gpu = str(device).startswith("cuda")
        self.streams = _GpuStreams(server, torch, device) if on_gpu else _NoStreams()
        self.ev_r = [torch.cuda.Event(), torch.cuda.Event()] if on_gpu else None
        self.step = 0
        self.n_reply_cts = self.per * reply_cts

    def submit(self) -> None:
        srv, st = self.server, self.streams
        srv.check_ready()
        b = self.step & 1
        if self.ev_r is not None and self.step >= 2:
            # the reduce that last read partial[b] (two steps ago) is done before a lane writes there again.  The main
            # stream carries nothing else -- in particular it never joins the lanes -- so this fork does not make a
            # lane wait for the other lane's tail of the previous step: groups keep alternating over the lanes across
            # step boundaries exactly as in the single-GPU pipeline
            st.main.wait_event(self.ev_r[b])
            srv.fork()
        # the groups write their replies straight into partial[b] (no device-to-device copy of the batch)
        srv.batch_set_reply_buffer(self.partial[b].data_ptr(), self.partial[b].shape[0] * self.partial[b].shape[1])
        srv.run_batch()                             # every staged query on this rank's shard (asynchronous)
        srv.join_stream(st.comm_handle())           # the communication stream -- not the main one -- follows the lanes
        with st.comm():
            self.comm.reduce_scatter_sum(self._replies[b], self.partial[b], self.rank)
            srv.reduce_fixup_device_async(self._replies[b].data_ptr(), self.n_reply_cts, st.comm_handle())
            if self.ev_r is not None:
                self.ev_r[b].record(st.side)
        self.step += 1

    def flush(self) -> None:
        self.streams.synchronize()
        self.server.sync()

    def close(self) -> None:
        """Gives the context its own reply buffer back (the plain batch calls fetch from there)."""
        self.flush()
        self.server.batch_set_reply_buffer(0, 0)

    def replies(self, step: int):
        return self._replies[step & 1]


# ------------------------------------------------------------------------------------------------------------------
# Slot-sharded step (pirgpu_slots_*): every rank holds 1 / G of the NTT slots of EVERY plaintext.
# ------------------------------------------------------------------------------------------------------------------

def slot_cuts(k_n: int, world: int, align: int = 16) -> List[int]:
    """Slot ranges [cuts[r], cuts[r + 1]) of the ring's k * N NTT slots, contiguous, balanced, multiples of `align`
    (the packing kernels work on blocks of 16 slots)."""
    if k_n % align:
        raise ValueError("k * N must be a multiple of %d" % align)
    blocks = k_n // align
    return [(blocks * r // world) * align for r in range(world + 1)]


class SlotsBuffers:
    """Device buffers of one slot-sharded step for one (server, batch size, world)."""

    def __init__(self, server, batch: int, rank: int, world: int, torch, device):
        if batch % world:
            raise ValueError("batch size must be a multiple of the world size")
        self.per = batch // world
        self.groups = (self.per + GROUP - 1) // GROUP
        k, N = server.k, server.N
        self.k_n = k * N
        self.ctw = 2 * k * N
        self.cuts = slot_cuts(self.k_n, world)
        self.width = [self.cuts[r + 1] - self.cuts[r] for r in range(world)]
        self.mine = self.width[rank]
        # row sums per query and slot: the C side (pirgpu_slots_scan_async / _finish_async) strides them by 2 * scan_rows,
        # scan_rows = ceil(num_pt / dimensions[1]) -- equal to dimensions[0] for the reference's CalculateDimensions, smaller
        # when user-supplied dimensions leave the last rows empty; the exchange must be sized from the same number
        rows = server.scan_info()["rows"] if hasattr(server, "scan_info") else server.params.dimensions[0]
        self.rc = 2 * int(rows)
        dim_sum = sum(server.params.dimensions)
        reply_cts = server.db.reply_ct_count()
        piece = [server.slots_packed_bytes(w) for w in self.width]      # one group's packed column selectors, rank r's slots
        if not all(piece):
            raise ValueError("the slot-sharded step needs d = 2 and the int8-MFMA scan in one column chunk")
        self.piece = piece
        self.piece_mine = piece[rank]
        self.packed_send = torch.empty((self.groups * sum(piece),), dtype=torch.uint8, device=device)
        self.packed_recv = torch.empty((world * self.groups * piece[rank],), dtype=torch.uint8, device=device)
        self.sv = torch.empty((self.per, dim_sum, self.ctw), dtype=torch.int64, device=device)
        self.rows_send = torch.empty((batch * self.rc * self.mine,), dtype=torch.int64, device=device)
        self.rows_recv = torch.empty((self.per * self.rc * self.k_n,), dtype=torch.int64, device=device)
        self.replies = torch.empty((self.per, reply_cts, 2, k, N), dtype=torch.int64, device=device)
        self.x1_send = [self.groups * p for p in piece]
        self.x1_recv = [self.groups * piece[rank]] * world
        self.x2_send = [self.per * self.rc * self.mine] * world
        self.x2_recv = [self.per * self.rc * w for w in self.width]
        # row sums may cross the links in 5 bytes per residue (moduli below 2^40); off by default: at 8 GPUs the
        # exchange is not what bounds the step and the two packing passes cost compute (PIRGPU_SLOTS_PACK40=1)
        self.rows40 = (os.environ.get("PIRGPU_SLOTS_PACK40", "0") == "1" and hasattr(server, "pack40_supported")
                       and server.pack40_supported())
        if self.rows40:
            self.rows_send40 = torch.empty((self.rows_send.numel() * 5 // 4,), dtype=torch.int32, device=device)
            self.rows_recv40 = torch.empty((self.rows_recv.numel() * 5 // 4,), dtype=torch.int32, device=device)

    def exchange_bytes_per_query(self, world: int) -> float:
        """Bytes a rank receives per query of the batch: its slots of every other rank's packed column selectors +
        the row sums of its own queries from the other ranks' slots."""
        batch = self.per * world
        x1 = (world - 1) * self.groups * self.piece_mine
        x2 = self.per * self.rc * (self.k_n - self.mine) * (5 if self.rows40 else 8)
        return (x1 + x2) / batch

    def exchange_selectors(self, comm) -> None:
        comm.all_to_all(self.packed_recv, self.packed_send, self.x1_recv, self.x1_send, units=self.groups)

    def exchange_rowsums(self, comm, server, stream: int = 0) -> None:
        """X2.  stream = 0 is the SYNCHRONOUS form (run_batch_slots): the packing kernels run on the library's own stream,
        the collective on torch's current one, so the host orders them -- a wait after the pack (the collective must not
        read rows_send40 before it is written) and after the collective (the unpack must not read rows_recv40 before it
        has arrived).  With a stream handle (SlotsPipeline: the communication stream, which is also torch's current
        stream there) pack, collective and unpack are ordered by that stream."""
        if self.rows40:
            server.pack40_async(self.rows_send.data_ptr(), self.rows_send40.data_ptr(), self.rows_send.numel(), stream)
            if not stream:
                server.sync()
            comm.all_to_all(self.rows_recv40, self.rows_send40, [v * 5 // 4 for v in self.x2_recv],
                            [v * 5 // 4 for v in self.x2_send], units=self.per)
            if not stream:
                comm.wait(self.rows_recv40)
            server.unpack40_async(self.rows_recv40.data_ptr(), self.rows_recv.data_ptr(), self.rows_recv.numel(), stream)
        else:
            comm.all_to_all(self.rows_recv, self.rows_send, self.x2_recv, self.x2_send, units=self.per)


def slots_exchange_supported(server) -> bool:
    """True when this server can take part in the slot-sharded step (d = 2, int8-MFMA scan in one column chunk)."""
    return hasattr(server, "slots_packed_bytes") and server.slots_packed_bytes(16) > 0


def run_batch_slots(server, bufs: SlotsBuffers, dist, rank: int, world: int, comm: Optional[Comm] = None,
                    first: int = 0) -> dict:
    """One SYNCHRONOUS slot-sharded step over a staged batch on `world` GPUs (d = 2):

      E  every rank expands its own `per` queries (groups of 8), keeps their NTT-form selection vectors and packs their
         column selectors into the scan's B-operand layout, cut by destination rank;
      X1 all-to-all: a rank receives ITS slots of every query's packed column selectors (1 / world of each);
      S  every rank scans its slots of the WHOLE matrix for all queries of the step in one launch;
      X2 all-to-all: the row sums return to the rank that expanded the query;
      U  that rank assembles them and runs inverse NTT + upper level for its own queries with the row selectors it kept.

    Rank r ends with the finished replies of the queries it expanded in bufs.replies (reply i answers query i, reference
    server.cpp:60-63).  No row-selector exchange, no reduce.  Returns the serial phase times."""
    comm = comm or Comm(dist, world)
    server.check_ready()
    import time
    t = [time.perf_counter()]
    lo = first + rank * bufs.per
    server.slots_expand_async(lo, bufs.per, bufs.packed_send.data_ptr(), bufs.sv.data_ptr(), bufs.cuts)
    server.sync()
    t.append(time.perf_counter())
    bufs.exchange_selectors(comm)
    t.append(time.perf_counter())
    server.slots_scan_async(bufs.packed_recv.data_ptr(), world, bufs.per, bufs.rows_send.data_ptr())
    server.sync()
    t.append(time.perf_counter())
    bufs.exchange_rowsums(comm, server)
    server.sync()
    t.append(time.perf_counter())
    server.slots_finish_async(bufs.rows_recv.data_ptr(), bufs.per, bufs.sv.data_ptr(), bufs.cuts, bufs.replies.data_ptr())
    server.sync()
    t.append(time.perf_counter())
    return {"expand_ms": (t[1] - t[0]) * 1e3, "exchange_selectors_ms": (t[2] - t[1]) * 1e3, "scan_ms": (t[3] - t[2]) * 1e3,
            "exchange_rowsums_ms": (t[4] - t[3]) * 1e3, "finish_ms": (t[5] - t[4]) * 1e3}


class _SlotStreamsCpu:
    """No streams on the CPU (oracle-backed server in the gloo tests): everything is synchronous."""

    def comm(self):
        return _NoStreams._Ctx()

    def gate_after(self, *events):
        return 0

    def record(self, name, b):
        pass

    def comm_wait(self, name, b):
        pass

    def comm_handle(self):
        return 0

    def fin_handle(self):
        return 0

    def synchronize(self):
        pass


class _SlotStreamsGpu:
    """The communication stream, a `fin` stream the finished steps report to, and a gate stream through which a lane is
    made to wait for exactly the events it needs (pirgpu_slots_*'s `after` takes a stream's current position)."""

    def __init__(self, torch, device, n_sets):
        self.torch = torch
        self.side = torch.cuda.Stream(device=device)
        self.fin = torch.cuda.Stream(device=device)
        self.gate = torch.cuda.Stream(device=device)
        self.ev = {name: [torch.cuda.Event() for _ in range(n_sets)] for name in ("x1", "x2", "u")}
        self.recorded = {name: [False] * n_sets for name in self.ev}

    def comm(self):
        return self.torch.cuda.stream(self.side)

    def gate_after(self, *events):
        """Handle of the gate stream after it has been made to wait for the named events (those recorded so far)."""
        for name, b in events:
            if self.recorded[name][b]:
                self.gate.wait_event(self.ev[name][b])
        return int(self.gate.cuda_stream)

    def record(self, name, b):
        self.ev[name][b].record(self.fin if name == "u" else self.side)
        self.recorded[name][b] = True

    def comm_wait(self, name, b):
        if self.recorded[name][b]:
            self.side.wait_event(self.ev[name][b])

    def comm_handle(self):
        return int(self.side.cuda_stream)

    def fin_handle(self):
        return int(self.fin.cuda_stream)

    def synchronize(self):
        self.side.synchronize()
        self.fin.synchronize()
        self.gate.synchronize()


class SlotsPipeline:
    """The slot-sharded step of run_batch_slots, PIPELINED over consecutive steps and free of host waits.  What submit(s)
    queues, in this order:

        lane:   S_(s-1)                    E_s                    U_(s-2)
        comm:            X2_(s-1)                   X1_s

    so both exchanges of a step run under the compute of its neighbours: E (VALU-bound) shares the chip with S
    (HBM-bound) like the two lanes of the single-GPU pipeline.  Three buffer sets (the selection vectors of step s are
    read by U_s two submits later).  Every edge is an event: a lane waits for exactly the exchange it consumes
    (pirgpu_slots_*'s `after`), the communication stream waits for exactly the lane that produced what it sends
    (`then`); nothing in submit() blocks the host.  The replies of step s are in `replies(s)` after flush() (or once
    submit(s + 2) has been followed by a synchronise of the `fin` stream) and are overwritten by submit(s + 5).

    submit(first) serves the staged queries [first + rank * per, first + (rank + 1) * per)."""

    SETS = 3

    def __init__(self, server, batch: int, rank: int, world: int, dist, torch, device, comm: Optional[Comm] = None):
        self.server, self.rank, self.world = server, rank, world
        self.sets = [SlotsBuffers(server, batch, rank, world, torch, device) for _ in range(self.SETS)]
        self.per = self.sets[0].per
        on_gpu = str(device).startswith("cuda")
        self.comm = comm or Comm(dist, world, host_sync=False)
        self.comm.host_sync = False
        self.streams = _SlotStreamsGpu(torch, device, self.SETS) if on_gpu else _SlotStreamsCpu()
        self.step = 0            # steps submitted (E + X1 queued)
        self.scanned = 0         # steps whose S + X2 are queued
        self.finished = 0        # steps whose U is queued

    def _scan(self, s: int) -> None:
        """S_s + X2_s."""
        srv, st, bufs = self.server, self.streams, self.sets[s % self.SETS]
        b = s % self.SETS
        srv.slots_scan_async(bufs.packed_recv.data_ptr(), self.world, bufs.per, bufs.rows_send.data_ptr(),
                             after=st.gate_after(("x1", b)), then=st.comm_handle())
        st.comm_wait("u", b)                   # the finish that last read rows_recv of this set (three steps ago)
        with st.comm():
            bufs.exchange_rowsums(self.comm, srv, st.comm_handle())
            st.record("x2", b)

    def _finish(self, s: int) -> None:
        """U_s."""
        srv, st, bufs = self.server, self.streams, self.sets[s % self.SETS]
        b = s % self.SETS
        srv.slots_finish_async(bufs.rows_recv.data_ptr(), bufs.per, bufs.sv.data_ptr(), bufs.cuts, bufs.replies.data_ptr(),
                               after=st.gate_after(("x2", b)), then=st.fin_handle())
        st.record("u", b)

    def submit(self, first: int = 0) -> None:
        srv, st = self.server, self.streams
        srv.check_ready()
        s = self.step
        b = s % self.SETS
        bufs = self.sets[b]
        while self.scanned < s:
            self._scan(self.scanned)
            self.scanned += 1
        # E_s refills set b: the exchange that read its send buffer and the finish that read its selection vectors
        # (both three steps ago) are done
        srv.slots_expand_async(first + self.rank * bufs.per, bufs.per, bufs.packed_send.data_ptr(), bufs.sv.data_ptr(),
                               bufs.cuts, after=st.gate_after(("x1", b), ("u", b)), then=st.comm_handle())
        with st.comm():
            bufs.exchange_selectors(self.comm)
            st.record("x1", b)
        while self.finished + 2 <= s:
            self._finish(self.finished)
            self.finished += 1
        self.step += 1

    def flush(self) -> None:
        """Queues what is still owed for the last two steps and waits for everything."""
        s = self.step
        while self.scanned < s:
            self._scan(self.scanned)
            self.scanned += 1
        while self.finished < s:
            self._finish(self.finished)
            self.finished += 1
        self.streams.synchronize()
        self.server.sync()

    def replies(self, step: int):
        """The replies tensor of `step` (0-based submit index): [per, reply_cts, 2, k, N] int64."""
        return self.sets[step % self.SETS].replies
