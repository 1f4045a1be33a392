"""World-size-2 gloo tests of the multi-GPU glue on CPU.

`pir_amd.distributed` is the product code under test: shard ranges, the packed exchange step
(run_batch_rows_packed: all-gather of packed column selectors, all-to-all of row selectors, reduce-scatter of the
partial replies, mod-q fix-up) and the whole-selection-vector step (run_batch_query_parallel: all-gather + all-reduce).
Two processes run them over gloo with a stand-in for the GPU server that is backed by the CPU oracle (it implements
the same duck-typed methods on host tensors; its "packed" format is simply the raw column selectors -- the glue only
moves bytes), and every rank checks the replies it ends up with against the oracle's full-database replies."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pir_amd import distributed as D
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
            assert D.row_cuts(n, world) == [c[0] for c in cuts] + [n]
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)
    with pytest.raises(ValueError):
        D.owned_queries(5, 0, 2)
    D.check_sum_fits((1 << 61) - 1, 8)
    with pytest.raises(ValueError):
        D.check_sum_fits((1 << 61) - 1, 9)


def test_slot_cuts_partition_the_ring():
    """Slot shards: contiguous, balanced to within one block of 16 slots, multiples of 16, covering [0, k N)."""
    for k_n in (2 * 2048, 2 * 4096, 3 * 8192, 4 * 16384, 4 * 8192):
        for world in (1, 2, 3, 4, 5, 8, 16):
            cuts = D.slot_cuts(k_n, world)
            assert len(cuts) == world + 1 and cuts[0] == 0 and cuts[-1] == k_n
            sizes = [b - a for a, b in zip(cuts[:-1], cuts[1:])]
            assert all(c % 16 == 0 for c in cuts) and min(sizes) > 0 and max(sizes) - min(sizes) <= 16
    with pytest.raises(ValueError):
        D.slot_cuts(8200, 2)


def test_slots_buffers_split_sizes_add_up():
    """The four split lists of the two all-to-alls cover their buffers exactly, for even and uneven cuts and partial
    groups (sizes only: a stand-in server that answers the geometry questions)."""
    class _Srv:
        k, N = 2, 4096

        class params:
            dimensions = [18, 21]

        class db:
            @staticmethod
            def reply_ct_count():
                return 8

        @staticmethod
        def slots_packed_bytes(slots):
            return slots * 2 * 1152          # KG = 2 column groups, 1152 bytes per tile set

    for world, batch in ((1, 3), (2, 18), (3, 9), (8, 72), (8, 8)):
        for rank in range(world):
            b = D.SlotsBuffers(_Srv, batch, rank, world, torch, "cpu")
            assert sum(b.x1_send) == b.packed_send.numel() and sum(b.x1_recv) == b.packed_recv.numel()
            assert sum(b.x2_send) == b.rows_send.numel() and sum(b.x2_recv) == b.rows_recv.numel()
            assert b.groups == (b.per + D.GROUP - 1) // D.GROUP and b.per * world == batch
            assert b.x1_recv == [b.groups * b.piece[rank]] * world and b.x2_send == [b.per * b.rc * b.mine] * world
            if world > 1:
                assert 0 < b.exchange_bytes_per_query(world) < 2 * 8192 * (18 + 21) * 8


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _view(ptr, n, dtype=np.uint64):
    ct = {np.uint64: C.c_uint64, np.uint8: C.c_uint8}[dtype]
    return np.ctypeslib.as_array((ct * n).from_address(ptr))


def _view32(ptr, n):
    return np.ctypeslib.as_array((C.c_uint32 * n).from_address(ptr))


class _Db:
    def __init__(self, n):
        self._n = n

    def reply_ct_count(self):
        return self._n


class OracleShardServer:
    """The methods pir_amd.distributed calls on a server, on host memory, computed by the CPU oracle."""

    def __init__(self, setup, rank, world):
        self.s, self.params = setup, setup.params
        self.orc = setup.orc
        self.k, self.N = self.orc.k, self.orc.N
        self.ctw = 2 * self.k * self.N
        self.dims = list(self.params.dimensions)
        self.lo, self.hi = shard_range(self.dims[0], rank, world)
        stride = int(np.prod(self.dims[1:])) if len(self.dims) > 1 else 1
        self.shard_db = setup.db_ntt[self.lo * stride:min(self.hi * stride, self.params.num_pt)]
        self.db = _Db(self.orc.reply_ct_count(len(self.dims)))
        self.queries, self.partials = None, None
        if len(self.dims) == 2:
            self._slot_setup(rank, world)

    # -- staging / expansion -------------------------------------------------------------------------------
    def stage_batch(self, queries):
        self.queries = queries

    def _sv(self, i):
        rc, sv = self.orc.oblivious_expansion_multi(self.queries[i], self.params.dim_sum, self.s.galois_keys)
        assert rc == 0
        return sv

    def packed_selector_bytes(self):
        return D.GROUP * self.dims[1] * self.ctw * 8 if len(self.dims) == 2 and self.hi > self.lo else 0

    def batch_expand_packed(self, first, count, packed_ptr, rows_ptr, cuts):
        n0, n1 = self.dims
        groups = (count + D.GROUP - 1) // D.GROUP
        packed = _view(packed_ptr, groups * D.GROUP * n1 * self.ctw).reshape(groups, D.GROUP, n1, self.ctw)
        rows = _view(rows_ptr, count * n0 * self.ctw)
        for i in range(count):
            sv = self._sv(first + i).reshape(n0 + n1, self.ctw)
            packed[i // D.GROUP, i % D.GROUP] = sv[n0:]
            for s in range(len(cuts) - 1):
                r0, nr = cuts[s], cuts[s + 1] - cuts[s]
                off = (count * r0 + i * nr) * self.ctw
                rows[off:off + nr * self.ctw] = sv[r0:r0 + nr].reshape(-1)

    def batch_expand(self, first, count, dst_ptr):
        out = _view(dst_ptr, count * self.params.dim_sum * self.ctw).reshape(count, -1)
        for i in range(count):
            out[i] = self._sv(first + i).reshape(-1)

    # -- multiply on the shard -------------------------------------------------------------------------------
    def _partial(self, rows_sel, rest_sel):
        sub_dims = [self.hi - self.lo] + self.dims[1:]
        sv = np.concatenate([rows_sel, rest_sel]).reshape(-1, 2, self.k, self.N).copy()
        rc, part = self.orc.db_multiply(np.ascontiguousarray(self.shard_db), sub_dims, sv)
        assert rc == 0
        return part

    def batch_run_packed(self, packed_ptr, n_ranks, per_rank, rows_ptr):
        n1, my = self.dims[1], self.hi - self.lo
        groups = (per_rank + D.GROUP - 1) // D.GROUP
        packed = _view(packed_ptr, n_ranks * groups * D.GROUP * n1 * self.ctw).reshape(n_ranks, groups, D.GROUP, n1, self.ctw)
        rows = _view(rows_ptr, n_ranks * per_rank * my * self.ctw).reshape(n_ranks * per_rank, my, self.ctw)
        self.partials = [self._partial(rows[r * per_rank + i], packed[r, i // D.GROUP, i % D.GROUP])
                         for r in range(n_ranks) for i in range(per_rank)]

    reply_buffer = None   # pirgpu_batch_set_reply_buffer: (pointer, capacity in ciphertexts)

    def batch_set_reply_buffer(self, ptr, cap_cts):
        self.reply_buffer = (ptr, cap_cts) if ptr else None

    def run_batch(self):
        """The plain batch pipeline on this shard: every staged query expanded here, multiplied against the shard."""
        self._batch_count = len(self.queries)
        self.partials = []
        for i in range(len(self.queries)):
            sv = self._sv(i).reshape(self.params.dim_sum, self.ctw)
            self.partials.append(self._partial(sv[self.lo:self.hi], sv[self.dims[0]:]) if len(self.dims) > 1
                                 else self._partial(sv[self.lo:self.hi], sv[:0]))
        if self.reply_buffer:
            ptr, cap = self.reply_buffer
            assert cap >= len(self.partials) * self.db.reply_ct_count()
            self.batch_reply_copy_to_device(ptr)

    def batch_run_selectors(self, sv_ptr, count):
        ds = self.params.dim_sum
        sv = _view(sv_ptr, count * ds * self.ctw).reshape(count, ds, self.ctw)
        self.partials = [self._partial(sv[i, self.lo:self.hi], sv[i, self.dims[0]:]) for i in range(count)]

    def batch_reply_copy_to_device(self, dst_ptr):
        out = _view(dst_ptr, len(self.partials) * self.db.reply_ct_count() * self.ctw)
        out[:] = np.stack(self.partials).reshape(-1)

    def reduce_fixup_device_n(self, ptr, n_cts):
        a = _view(ptr, n_cts * self.ctw).reshape(n_cts, 2, self.k, self.N)
        for j, qj in enumerate(self.orc.moduli[: self.k]):
            a[:, :, j, :] %= np.uint64(qj)

    # -- the pipelined step's entry points: no streams on the CPU, so ordering calls are no-ops -----------------
    def stream_handle(self):
        return 0

    def fork(self):
        pass

    def join(self):
        pass

    def join_stream(self, stream):
        pass

    # -- row selectors in 5 bytes per residue (pirgpu_pack40_device_async / _unpack40_): 4 words <-> 5 dwords -----------
    def pack40_supported(self):
        return all(int(q) < 2 ** 40 for q in self.orc.moduli[: self.k])

    def pack40_async(self, words_ptr, packed_ptr, words, stream=0):
        w = _view(words_ptr, words).reshape(-1, 4)
        out = _view32(packed_ptr, words * 5 // 4).reshape(-1, 5)
        out[:, :4] = (w & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        hi = (w >> np.uint64(32)).astype(np.uint32)
        assert (hi < 256).all()
        out[:, 4] = hi[:, 0] | (hi[:, 1] << 8) | (hi[:, 2] << 16) | (hi[:, 3] << 24)

    def unpack40_async(self, packed_ptr, words_ptr, words, stream=0):
        p = _view32(packed_ptr, words * 5 // 4).reshape(-1, 5)
        w = _view(words_ptr, words).reshape(-1, 4)
        for i in range(4):
            w[:, i] = p[:, i].astype(np.uint64) | (((p[:, 4] >> (8 * i)) & 0xFF).astype(np.uint64) << np.uint64(32))

    def sync(self):
        pass

    def batch_expand_packed_async(self, first, count, packed_ptr, rows_ptr, cuts):
        self.batch_expand_packed(first, count, packed_ptr, rows_ptr, cuts)

    def batch_reply_copy_to_device_async(self, dst_ptr):
        self.batch_reply_copy_to_device(dst_ptr)

    def reduce_fixup_device_async(self, ptr, n_cts, stream=0):
        self.reduce_fixup_device_n(ptr, n_cts)

    # -- slot-sharded step (pirgpu_slots_*): this stand-in holds the slots [cuts[rank], cuts[rank+1]) of every plaintext ----
    # its "packed" format is the raw NTT-form column selectors of a group restricted to a slot range:
    # u64 [GROUP][n1][2][slots]; row sums and selection vectors are in the oracle's own slot order
    def _slot_setup(self, rank, world):
        self.rank, self.world = rank, world
        self.kN = self.k * self.N
        self.slot_cut = D.slot_cuts(self.kN, world)
        self.c0, self.c1 = self.slot_cut[rank], self.slot_cut[rank + 1]
        n0, n1 = self.dims
        db = np.zeros((n0 * n1, self.kN), dtype=np.uint64)
        db[: self.params.num_pt] = np.asarray(self.s.db_ntt).reshape(self.params.num_pt, self.kN)
        self.db_slots = db[:, self.c0:self.c1].reshape(n0, n1, self.c1 - self.c0).copy()   # ONLY this rank's slots
        self.qvec = np.repeat(np.array(self.orc.moduli[: self.k], dtype=np.uint64), self.N)

    @staticmethod
    def _mulmod(a, b, q):
        """a * b mod q elementwise in uint64 arithmetic (operands < q < 2^42)."""
        assert int(q.max()) < 1 << 42
        a0, a1 = a & np.uint64((1 << 21) - 1), a >> np.uint64(21)
        hi = (a1 * b) % q
        return ((hi << np.uint64(21)) % q + (a0 * b) % q) % q

    def slots_packed_bytes(self, slots):
        return D.GROUP * self.dims[1] * 2 * slots * 8 if len(self.dims) == 2 else 0

    def _sv_ntt(self, i):
        sv = self._sv(i)                                     # coefficient form [dim_sum, 2, k, N]
        return np.stack([self.orc.ct_ntt_fwd(ct) for ct in sv])

    def slots_expand_async(self, first, count, packed_ptr, sv_ptr, cuts, after=0, then=0):
        n0, n1 = self.dims
        G = len(cuts) - 1
        groups = (count + D.GROUP - 1) // D.GROUP
        sv_out = _view(sv_ptr, count * (n0 + n1) * self.ctw).reshape(count, n0 + n1, self.ctw)
        total = groups * D.GROUP * n1 * 2 * self.kN
        packed = _view(packed_ptr, total)
        off = [groups * D.GROUP * n1 * 2 * cuts[r] for r in range(G)]        # piece [rank r][group g]
        for i in range(count):
            sv = self._sv_ntt(first + i).reshape(n0 + n1, 2, self.kN)
            sv_out[i] = sv.reshape(n0 + n1, self.ctw)
            g, q = divmod(i, D.GROUP)
            for r in range(G):
                w = cuts[r + 1] - cuts[r]
                piece = packed[off[r] + g * D.GROUP * n1 * 2 * w: off[r] + (g + 1) * D.GROUP * n1 * 2 * w]
                piece.reshape(D.GROUP, n1, 2, w)[q] = sv[n0:, :, cuts[r]:cuts[r + 1]]

    def slots_scan_async(self, packed_ptr, n_ranks, per_rank, rowsums_ptr, after=0, then=0):
        n0, n1 = self.dims
        w = self.c1 - self.c0
        groups = (per_rank + D.GROUP - 1) // D.GROUP
        packed = _view(packed_ptr, n_ranks * groups * D.GROUP * n1 * 2 * w).reshape(n_ranks, groups, D.GROUP, n1, 2, w)
        out = _view(rowsums_ptr, n_ranks * per_rank * n0 * 2 * w).reshape(n_ranks, per_rank, n0, 2, w)
        qv = self.qvec[self.c0:self.c1]
        for r in range(n_ranks):
            for i in range(per_rank):
                sel = packed[r, i // D.GROUP, i % D.GROUP]             # [n1][2][w]
                acc = np.zeros((n0, 2, w), dtype=np.uint64)
                for col in range(n1):
                    for comp in range(2):
                        acc[:, comp] = (acc[:, comp] + self._mulmod(self.db_slots[:, col], sel[col, comp][None, :], qv[None, :])) % qv
                out[r, i] = acc

    def slots_finish_async(self, rowsums_ptr, count, sv_ptr, cuts, replies_ptr, after=0, then=0):
        """database.cpp:196-254 on assembled row sums, built from the oracle's primitives."""
        n0, n1 = self.dims
        G = len(cuts) - 1
        src = _view(rowsums_ptr, count * n0 * 2 * self.kN)
        sv_all = _view(sv_ptr, count * (n0 + n1) * self.ctw).reshape(count, n0 + n1, 2, self.k, self.N)
        E = 2 * self.orc.expansion_ratio()
        out = _view(replies_ptr, count * E * self.ctw).reshape(count, E, 2, self.k, self.N)
        rows = np.empty((count, n0, 2, self.kN), dtype=np.uint64)
        for h in range(G):
            w = cuts[h + 1] - cuts[h]
            blk = src[count * n0 * 2 * cuts[h]: count * n0 * 2 * cuts[h + 1]].reshape(count, n0, 2, w)
            rows[:, :, :, cuts[h]:cuts[h + 1]] = blk
        qk = np.array(self.orc.moduli[: self.k], dtype=np.uint64)[None, :, None]
        for i in range(count):
            result = np.zeros((E, 2, self.k, self.N), dtype=np.uint64)
            for r in range(n0):
                ct = self.orc.ct_ntt_inv(rows[i, r].reshape(2, self.k, self.N).copy())
                pts = self.orc.reencode(ct)                           # [E][N] coefficients below 2^b
                sel = np.ascontiguousarray(sv_all[i, r])
                for e in range(E):
                    temp = self.orc.multiply_plain_ntt(sel, self.orc.plain_lift_ntt(pts[e]))
                    result[e] = (result[e] + temp) % qk
            out[i] = np.stack([self.orc.ct_ntt_inv(result[e].copy()) for e in range(E)])

    # -- the collective transparent-ciphertext decision (pirgpu_zero_plaintexts / _set_remote_ / _check_ready) ------
    remote_zero = 0

    def zero_plaintexts(self):
        return int(sum(1 for pt in self.shard_db if not pt.any()))

    def set_remote_zero_plaintexts(self, n):
        self.remote_zero = int(n)

    def check_ready(self):
        if self.zero_plaintexts() + self.remote_zero:
            from pir_amd.server import PirGpuError
            raise PirGpuError(13, "result ciphertext is transparent")


def rows_step_check(rank, world, d, dbsize, elem, batch, zero_pt=None):
    """One rank of the row-sharded step over an initialised gloo process group with the oracle-backed server: True when
    every reply this rank ends up with equals the oracle's full-database reply (and decodes to the item).  zero_pt:
    index of a plaintext made identically zero -- every rank must then fail with the reference's status, none hangs."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from pir_fixtures import PirSetup
    from pir_amd.server import PirGpuError
    s = PirSetup(dbsize, elem, d, N=4096, plain_bits=24)      # same seeds on every rank
    p = s.params
    if zero_pt is not None:
        s.db_ntt[zero_pt] = 0
    indexes = [(dbsize - 2 - 37 * i) % dbsize for i in range(batch)]
    queries = [s.client.create_query_for(p, i) for i in indexes]
    srv = OracleShardServer(s, rank, world)
    srv.stage_batch(queries)
    comm = D.Comm(dist, world)
    assert not comm.device_native
    D.check_sum_fits(max(s.orc.moduli[: s.orc.k]), world)
    total_zero = D.sync_zero_plaintexts(srv, dist, world, comm, torch, "cpu")
    ok = True
    lo, hi = D.owned_queries(batch, rank, world)
    if zero_pt is not None:
        # the reference fails EVERY query (database.cpp:313-315); here every rank -- also the one whose shard has no
        # zero plaintext -- raises before it enters a collective
        ok &= total_zero == 1
        bufs = D.PackedBuffers(srv, batch, rank, world, torch, "cpu")
        for step in (lambda: D.run_batch_rows_packed(srv, bufs, dist, rank, world, comm),
                     lambda: D.run_batch_query_parallel(srv, None, None, dist, rank, world, comm)):
            try:
                step()
                ok = False
            except PirGpuError as e:
                ok &= e.code == 13 and "transparent" in e.message
        return ok
    full = [s.orc.process_query(s.db_ntt, p.dimensions, queries[i], s.galois_keys)[1] for i in range(batch)]
    if d == 2:
        assert D.packed_exchange_supported(srv, dist, world, comm, torch, "cpu")
        bufs = D.PackedBuffers(srv, batch, rank, world, torch, "cpu")
        ok &= bufs.rows40                                      # 36-bit moduli: row selectors cross in 5 bytes
        D.run_batch_rows_packed(srv, bufs, dist, rank, world, comm)
        mine = bufs.replies.numpy().view(np.uint64)
        for i in range(lo, hi):                                # rank r ends with the replies of ITS queries
            ok &= bool(np.array_equal(mine[i - lo], full[i]))
            ok &= s.client.process_response(p, indexes[i], mine[i - lo]) == s.item(indexes[i])
        os.environ["PIRGPU_ROWS_PACK40"] = "0"                 # ... and the same step with u64 row selectors
        try:
            bufs8 = D.PackedBuffers(srv, batch, rank, world, torch, "cpu")
        finally:
            del os.environ["PIRGPU_ROWS_PACK40"]
        ok &= not bufs8.rows40 and bufs8.exchange_bytes_per_query(world) > bufs.exchange_bytes_per_query(world)
        D.run_batch_rows_packed(srv, bufs8, dist, rank, world, comm)
        ok &= bool(np.array_equal(bufs8.replies.numpy(), bufs.replies.numpy()))
        # the PIPELINED step (RowsPipeline): three consecutive steps over different queries -- step t serves the staged
        # queries [t * batch, (t + 1) * batch); the multiply + reduce of step t are queued by submit t + 1 (or flush),
        # the two buffer sets alternate, every rank checks the replies of ITS queries of every step
        steps = 3
        idx_all = indexes + [(dbsize - 5 - 53 * i) % dbsize for i in range(batch * (steps - 1))]
        q_all = queries + [s.client.create_query_for(p, i) for i in idx_all[batch:]]
        srv.stage_batch(q_all)
        pipe = D.RowsPipeline(srv, batch, rank, world, dist, torch, "cpu", comm=D.Comm(dist, world))
        seen = {}
        for t in range(steps):
            pipe.submit(first=t * batch)
            if t >= 1:                                         # step t - 1 is complete (CPU: everything is synchronous)
                seen[t - 1] = pipe.replies(t - 1).numpy().view(np.uint64).copy()
        pipe.flush()
        seen[steps - 1] = pipe.replies(steps - 1).numpy().view(np.uint64).copy()
        for t in range(steps):
            for i in range(lo, hi):
                g = t * batch + i
                want = full[i] if t == 0 else s.orc.process_query(s.db_ntt, p.dimensions, q_all[g], s.galois_keys)[1]
                ok &= bool(np.array_equal(seen[t][i - lo], want))
                ok &= s.client.process_response(p, idx_all[g], seen[t][i - lo]) == s.item(idx_all[g])
        # hybrid layout, degenerate at two ranks: 2 replica groups of ONE shard each -- every rank holds the whole
        # database, runs the pipelined step inside its own one-rank process group and serves its half of the queries
        gi, gr, S, groups = D.hybrid_layout(rank, world, 2)
        pgs = [dist.new_group(g, backend="gloo") for g in groups]
        whole = OracleShardServer(s, 0, 1)
        whole.stage_batch(q_all)
        bpg = batch // 2
        hp = D.RowsPipeline(whole, bpg, gr, S, dist, torch, "cpu", comm=D.Comm(dist, S, group=pgs[gi]))
        hp.submit(first=gi * bpg)
        hp.submit(first=batch + gi * bpg)
        hp.flush()
        for t, base in ((0, 0), (1, batch)):
            mine_h = hp.replies(t).numpy().view(np.uint64)
            for i in range(bpg):
                g = base + gi * bpg + i
                want = s.orc.process_query(s.db_ntt, p.dimensions, q_all[g], s.galois_keys)[1]
                ok &= bool(np.array_equal(mine_h[i], want))
        # replicated expansion (what bench.py uses at two GPUs): every rank expands every query itself, the only
        # collective is the reduce-scatter of the partial replies; two steps (both buffer sets)
        srv.stage_batch(queries)
        rp = D.RowsReplicatedPipeline(srv, batch, rank, world, dist, torch, "cpu", comm=D.Comm(dist, world))
        rp.submit()
        rp.submit()
        rp.close()
        for t in range(2):
            mine_r = rp.replies(t).numpy().view(np.uint64)
            for i in range(lo, hi):
                ok &= bool(np.array_equal(mine_r[i - lo], full[i]))
        srv.stage_batch(queries)
        # the SLOT-sharded step: every rank holds half of the NTT slots of every plaintext, receives its slots of every
        # query's column selectors, returns row sums to the query's owner, which finishes its own queries -- synchronous,
        # then pipelined over four steps (three buffer sets)
        assert D.slots_exchange_supported(srv)
        sb = D.SlotsBuffers(srv, batch, rank, world, torch, "cpu")
        D.run_batch_slots(srv, sb, dist, rank, world, comm)
        mine_s = sb.replies.numpy().view(np.uint64)
        for i in range(lo, hi):
            ok &= bool(np.array_equal(mine_s[i - lo], full[i]))
            ok &= s.client.process_response(p, indexes[i], mine_s[i - lo]) == s.item(indexes[i])
        os.environ["PIRGPU_SLOTS_PACK40"] = "1"                # ... and with the row sums crossing in 5 bytes per residue
        try:
            sb40 = D.SlotsBuffers(srv, batch, rank, world, torch, "cpu")
        finally:
            del os.environ["PIRGPU_SLOTS_PACK40"]
        ok &= sb40.rows40 and not sb.rows40 and sb40.exchange_bytes_per_query(world) < sb.exchange_bytes_per_query(world)
        D.run_batch_slots(srv, sb40, dist, rank, world, comm)
        ok &= bool(np.array_equal(sb40.replies.numpy(), sb.replies.numpy()))
        srv.stage_batch(q_all + q_all[:batch])
        sp = D.SlotsPipeline(srv, batch, rank, world, dist, torch, "cpu", comm=D.Comm(dist, world))
        for t in range(4):
            sp.submit(first=t * batch)
        sp.flush()
        for t in (1, 2, 3):
            got = sp.replies(t).numpy().view(np.uint64)
            for i in range(lo, hi):
                g = (t * batch + i) % len(q_all)
                want = full[i] if g < batch else s.orc.process_query(s.db_ntt, p.dimensions, q_all[g], s.galois_keys)[1]
                ok &= bool(np.array_equal(got[i - lo], want))
        srv.stage_batch(queries)
    # the whole-selection-vector exchange (any d): every rank ends with every reply
    sv_all = torch.empty((batch, p.dim_sum, 2, s.orc.k, 4096), dtype=torch.int64)
    replies = torch.empty((batch, srv.db.reply_ct_count(), 2, s.orc.k, 4096), dtype=torch.int64)
    D.run_batch_query_parallel(srv, sv_all, replies, dist, rank, world, comm)
    allr = replies.numpy().view(np.uint64)
    for i in range(batch):
        ok &= bool(np.array_equal(allr[i], full[i]))
    return ok


def _worker(rank, world, port, d, dbsize, elem, batch, out_q, zero_pt=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out_q.put((rank, rows_step_check(rank, world, d, dbsize, elem, batch, zero_pt)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("d,dbsize,elem,batch,zero_pt", [(2, 300, 288, 4, None), (1, 120, 288, 2, None),
                                                         (2, 300, 288, 2, 1)],
                         ids=["d2", "d1", "d2-zero-plaintext-in-rank0"])
def test_row_sharded_step_over_gloo(d, dbsize, elem, batch, zero_pt):
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, d, dbsize, elem, batch, out_q, zero_pt)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = dict(out_q.get(timeout=5) for _ in range(2))
    assert got == {0: True, 1: True}
