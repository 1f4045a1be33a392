"""Host-side mirror of the reference's PIRDatabase / PIRServer (pir/cpp/database.h, server.h)
over the C ABI.  Method names, argument meaning and error codes follow the reference; SEAL
objects are replaced by numpy residue arrays in SEAL's layout:

    ciphertext  uint64[2, k, N]         Galois key  uint64[k, 2, k+1, N] (NTT form)

All arithmetic runs in libpirgpu.so on the GPU; nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import enum
from typing import Dict, Iterable, List, Optional, Sequence, Union

import numpy as np

from . import capi
from .parameters import PIRParameters, calculate_dimensions


class StatusCode(enum.IntEnum):
    """absl::StatusCode values the reference returns on this path."""
    OK = 0
    INVALID_ARGUMENT = 3
    FAILED_PRECONDITION = 9
    UNIMPLEMENTED = 12
    INTERNAL = 13


class PirGpuError(Exception):
    def __init__(self, code: int, message: str):
        super().__init__("%s: %s" % (StatusCode(code).name if code in StatusCode._value2member_map_ else code,
                                     message))
        self.code = code
        self.message = message


def _u64(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(capi.u64p)


class PIRDatabase:
    """reference database.h:37-133.  Owns the device context and the HBM-resident encoded database."""

    def __init__(self, params: PIRParameters, device: int = 0, shard: Optional[Sequence[int]] = None,
                 slots: Optional[Sequence[int]] = None):
        """shard: rows [begin, end) of dimension 0 this context holds; slots: NTT slots [begin, end) of every plaintext
        it holds (multi-GPU partitionings; default: the whole database)."""
        self.params = params
        enc = params.encryption_parameters
        self.N = enc.poly_modulus_degree
        self.k = len(enc.coeff_modulus) - 1
        self.lib = capi.load()
        p = capi.make_params(params, device=device, shard=shard, slots=slots)
        self._cparams = p
        h = C.c_void_p()
        rc = self.lib.pirgpu_create(C.byref(p), C.byref(h))
        if rc != 0:
            raise PirGpuError(rc, self.lib.pirgpu_create_error().decode())
        self._h = h

    # -- lifetime ---------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self.lib.pirgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise PirGpuError(rc, self.lib.pirgpu_last_error(self._h).decode())

    @property
    def handle(self):
        return self._h

    # -- reference interface ------------------------------------------------------
    @classmethod
    def Create(cls, params: PIRParameters, rawdb=None, device: int = 0, shard=None, slots=None) -> "PIRDatabase":
        """database.cpp:40-58: Create(params) / Create(rawdb, params)."""
        db = cls(params, device=device, shard=shard, slots=slots)
        if rawdb is not None:
            db.populate(rawdb)
        return db

    def populate(self, rawdb) -> None:
        """database.cpp:84-110 (strings/bytes) -- encoding, plain lift and NTT run on the GPU."""
        if isinstance(rawdb, np.ndarray) and rawdb.dtype == np.uint8 and rawdb.ndim == 2:
            n, width = rawdb.shape
            buf = np.ascontiguousarray(rawdb)
        else:
            items = list(rawdb)
            n = len(items)
            if n != self.params.num_items:
                raise PirGpuError(3, "Database size %d does not match params value %d" % (n, self.params.num_items))
            width = self.params.bytes_per_item
            if any(len(it) != width for it in items):
                raise PirGpuError(3, "item size does not match parameters")
            buf = np.frombuffer(b"".join(items), dtype=np.uint8)
        self._check(self.lib.pirgpu_db_load_items(self._h, buf.ctypes.data_as(capi.u8p), n, width))

    def populate_coeffs(self, coeffs, first_pt: int = 0) -> None:
        """Plaintexts given as coefficient rows (< t), e.g. IntegerEncoder output (database.cpp:60-82)."""
        arr = np.zeros((len(coeffs), self.N), dtype=np.uint64)
        for i, row in enumerate(coeffs):
            row = np.asarray(row, dtype=np.uint64)
            arr[i, : row.shape[0]] = row
        self._check(self.lib.pirgpu_db_load_coeffs(self._h, first_pt, arr.shape[0], _ptr(arr)))

    def size(self) -> int:
        return int(self.lib.pirgpu_db_size(self._h))

    def set_transparent_policy(self, allow: bool) -> None:
        """False (default): an identically-zero database plaintext makes every query fail with Internal, like SEAL's
        "result ciphertext is transparent" through database.cpp:313-315; True: return the defined reply."""
        self._check(self.lib.pirgpu_set_transparent_policy(self._h, 1 if allow else 0))

    def set_option(self, name: str, value: int) -> None:
        """pirgpu_set_option: a tuning / behaviour option by name (before the first query for workspace-shaping ones)."""
        self._check(self.lib.pirgpu_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_int64(0)
        self._check(self.lib.pirgpu_get_option(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    def finalize(self, release_staging: bool = False) -> None:
        """Pack the operand-layout copy now; optionally free the u64 staging copy (no reloads afterwards)."""
        self._check(self.lib.pirgpu_db_finalize(self._h, 1 if release_staging else 0))

    def read_plaintext(self, index: int) -> np.ndarray:
        out = np.empty((self.k, self.N), dtype=np.uint64)
        self._check(self.lib.pirgpu_db_read_plaintext(self._h, index, _ptr(out)))
        return out

    def reply_ct_count(self) -> int:
        return int(self.lib.pirgpu_reply_ct_count(self._h))

    def expansion_ratio(self) -> int:
        return int(self.lib.pirgpu_expansion_ratio(self._h))

    def multiply(self, selection_vector) -> np.ndarray:
        """database.cpp:290-316: selection vector [dim_sum, 2, k, N] (coefficient form) -> reply cts."""
        sv = _u64(selection_vector)
        if sv.ndim != 4 or sv.shape[1:] != (2, self.k, self.N):
            raise PirGpuError(3, "selection vector must have shape [n, 2, %d, %d], got %s" % (self.k, self.N, list(sv.shape)))
        n = self.reply_ct_count()
        out = np.empty((n, 2, self.k, self.N), dtype=np.uint64)
        cnt = C.c_uint64(0)
        self._check(self.lib.pirgpu_multiply(self._h, _ptr(sv), sv.shape[0], _ptr(out), n, C.byref(cnt)))
        return out[: cnt.value]

    def calculate_indices(self, index: int) -> List[int]:
        return self.params.calculate_indices(index)

    def calculate_item_offset(self, index: int) -> int:
        return self.params.calculate_item_offset(index)

    calculate_dimensions = staticmethod(calculate_dimensions)


class PIRServer:
    """reference server.h:33-145."""

    def __init__(self, db: PIRDatabase, params: PIRParameters):
        self.db = db
        self.params = params
        self.lib = db.lib
        self.N, self.k = db.N, db.k

    @classmethod
    def Create(cls, db: PIRDatabase, params: PIRParameters) -> "PIRServer":
        """server.cpp:35-42"""
        full = db._cparams.shard_begin == 0 and db._cparams.shard_end in (0, params.dimensions[0])
        if full and params.num_pt != db.size():
            raise PirGpuError(3, "database size mismatch")
        return cls(db, params)

    def _check(self, rc):
        self.db._check(rc)

    def _cts(self, a, ndim: int, what: str) -> np.ndarray:
        """uint64, contiguous, shape [..., 2, k, N] with `ndim` dimensions -- checked before any pointer reaches C."""
        a = _u64(a)
        if a.ndim != ndim or a.shape[-3:] != (2, self.k, self.N):
            raise PirGpuError(3, "%s must have shape [%s2, %d, %d], got %s"
                              % (what, "n, " * (ndim - 3), self.k, self.N, list(a.shape)))
        return a

    # -- keys -----------------------------------------------------------------------
    def set_galois_keys(self, galois_keys: Dict[int, np.ndarray]) -> None:
        """Install what SEALDeserialize<GaloisKeys> yields per request (server.cpp:46-48)."""
        self._check(self.lib.pirgpu_clear_galois_keys(self.db.handle))
        for g, key in galois_keys.items():
            key = _u64(key)
            if key.shape != (self.k, 2, self.k + 1, self.N):
                raise PirGpuError(3, "Galois key must have shape [%d, 2, %d, %d], got %s"
                                  % (self.k, self.k + 1, self.N, list(key.shape)))
            self._check(self.lib.pirgpu_set_galois_key(self.db.handle, int(g), _ptr(key)))

    # -- per-client key sets (keys are per request in the reference, server.cpp:46-48) ------------
    def set_keyset_capacity(self, capacity: int) -> None:
        self._check(self.lib.pirgpu_set_keyset_capacity(self.db.handle, capacity))

    def install_keyset(self, client_id: bytes, galois_keys: Dict[int, np.ndarray]) -> int:
        """Makes one client's Galois keys resident under `client_id` (any bytes that identify the client; the wire
        layer uses the serialized GaloisKeys object) and returns the slot; a resident set is reused, not re-uploaded."""
        ident = np.frombuffer(client_id, dtype=np.uint8)
        slot = C.c_uint32(0)
        self._check(self.lib.pirgpu_keyset_lookup(self.db.handle, ident.ctypes.data_as(capi.u8p), len(client_id), 1,
                                                  C.byref(slot)))
        if slot.value:
            return int(slot.value)
        self._check(self.lib.pirgpu_keyset_claim(self.db.handle, ident.ctypes.data_as(capi.u8p), len(client_id),
                                                 C.byref(slot)))
        for g, key in galois_keys.items():
            key = _u64(key)
            if key.shape != (self.k, 2, self.k + 1, self.N):
                self.lib.pirgpu_keyset_release(self.db.handle, slot.value)
                raise PirGpuError(3, "Galois key must have shape [%d, 2, %d, %d], got %s"
                                  % (self.k, self.k + 1, self.N, list(key.shape)))
            self._check(self.lib.pirgpu_keyset_set_key(self.db.handle, slot.value, int(g), _ptr(key)))
        return int(slot.value)

    def release_keyset(self, slot: int) -> None:
        self._check(self.lib.pirgpu_keyset_release(self.db.handle, slot))

    def use_keyset(self, slot: int) -> None:
        """Key set of the single-query entry points (process_query, run_staged, oblivious_expansion, ...); 0 = the
        set installed with set_galois_keys."""
        self._check(self.lib.pirgpu_query_use_keyset(self.db.handle, slot))

    def set_batch_keysets(self, slots: Sequence[int]) -> None:
        """One key set slot per query of the staged batch (stage_batch resets them to 0)."""
        arr = (C.c_uint32 * len(slots))(*[int(x) for x in slots])
        self._check(self.lib.pirgpu_batch_set_keysets(self.db.handle, arr, len(slots)))

    def keyset_stats(self) -> Dict[str, int]:
        st = (C.c_uint64 * 4)()
        self._check(self.lib.pirgpu_keyset_stats(self.db.handle, st))
        return {"resident": int(st[0]), "key_uploads": int(st[1]), "evictions": int(st[2]), "capacity": int(st[3])}

    # -- query path -------------------------------------------------------------------
    def process_query(self, query, galois_keys: Optional[Dict[int, np.ndarray]] = None) -> np.ndarray:
        """processQuery (server.cpp:173-195) on residue arrays: query [nq, 2, k, N] -> reply cts."""
        if galois_keys is not None:
            self.set_galois_keys(galois_keys)
        q = self._cts(query, 4, "query")
        n = self.db.reply_ct_count()
        out = np.empty((n, 2, self.k, self.N), dtype=np.uint64)
        cnt = C.c_uint64(0)
        self._check(self.lib.pirgpu_process_query(self.db.handle, _ptr(q), q.shape[0], _ptr(out), n, C.byref(cnt)))
        return out[: cnt.value]

    def ProcessRequest(self, request: bytes) -> bytes:
        """server.cpp:44-65: serialized pir.Request -> serialized pir.Response."""
        buf = np.frombuffer(request, dtype=np.uint8)
        resp = C.c_void_p()
        rlen = C.c_size_t(0)
        self._check(self.lib.pirgpu_process_request(self.db.handle, buf.ctypes.data_as(capi.u8p), len(request),
                                                    C.byref(resp), C.byref(rlen)))
        try:
            return C.string_at(resp.value, rlen.value)
        finally:
            self.lib.pirgpu_free(resp)

    def ProcessRequests(self, requests: Sequence[bytes]):
        """n independent requests (different clients) served together: [(status, response bytes or error text)]."""
        n = len(requests)
        bufs = [np.frombuffer(r, dtype=np.uint8) for r in requests]
        ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
        lens = (C.c_size_t * n)(*[len(r) for r in requests])
        resp = (C.c_void_p * n)()
        rlen = (C.c_size_t * n)()
        status = (C.c_int * n)()
        self.lib.pirgpu_process_requests(self.db.handle, n, ptrs, lens, resp, rlen, status)
        out = []
        for i in range(n):
            if status[i] == 0:
                out.append((0, C.string_at(resp[i], rlen[i])))
                self.lib.pirgpu_free(resp[i])
            else:
                out.append((int(status[i]), None))
        self.request_errors = [self.lib.pirgpu_request_error(i).decode() for i in range(n)]   # "" where it succeeded
        return out

    class PendingRequests:
        """A call handed to the library by ProcessRequestsBegin.  The ctypes arrays below are written by the library's
        serving thread until the call has been ended; a token that is dropped without ProcessRequestsEnd ends it
        itself (close / __del__), and a second End on the same token is refused."""

        def __init__(self, lib, requests):
            n = len(requests)
            self.lib, self.n = lib, n
            self.requests = list(requests)
            self.bufs = [np.frombuffer(r, dtype=np.uint8) for r in self.requests]
            self.ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in self.bufs])
            self.lens = (C.c_size_t * n)(*[len(r) for r in self.requests])
            self.resp = (C.c_void_p * n)()
            self.rlen = (C.c_size_t * n)()
            self.status = (C.c_int * n)()
            self.call = C.c_void_p()

        def pending(self) -> bool:
            return bool(self.call)

        def end(self) -> int:
            """Waits for the call; returns the library's status code.  The token is spent afterwards."""
            call, self.call = self.call, C.c_void_p()
            return int(self.lib.pirgpu_process_requests_end(call))

        def close(self) -> None:
            if self.pending():
                self.end()
                for i in range(self.n):
                    if self.status[i] == 0 and self.resp[i]:
                        self.lib.pirgpu_free(self.resp[i])

        def __del__(self):
            try:
                self.close()
            except Exception:       # interpreter shutdown
                pass

    def ProcessRequestsBegin(self, requests: Sequence[bytes]):
        """ProcessRequests in two halves (pirgpu_process_requests_begin / _end): hands the call to a serving thread of
        the library and returns a token for ProcessRequestsEnd.  One calling thread keeps two calls in flight by calling
        Begin for the next batch of requests before End for the previous one."""
        tok = PIRServer.PendingRequests(self.lib, requests)
        self._check(self.lib.pirgpu_process_requests_begin(self.db.handle, tok.n, tok.ptrs, tok.lens, tok.resp, tok.rlen,
                                                           tok.status, C.byref(tok.call)))
        return tok

    def ProcessRequestsEnd(self, tok):
        """Waits for the call and returns [(status, response bytes or None)] like ProcessRequests."""
        if not tok.pending():
            raise PirGpuError(9, "ProcessRequestsEnd: this call has already been ended")   # FailedPrecondition
        rc = tok.end()
        # (the call's return value is the first non-zero per-request status: a failing request is reported in its own slot
        # below, like ProcessRequests does; only a failure of the call itself -- no request carries it -- is raised)
        if rc and not any(tok.status[i] for i in range(tok.n)):
            self._check(rc)
        out = []
        for i in range(tok.n):
            if tok.status[i] == 0:
                out.append((0, C.string_at(tok.resp[i], tok.rlen[i])))
                self.lib.pirgpu_free(tok.resp[i])
            else:
                out.append((int(tok.status[i]), None))
        self.request_errors = [self.lib.pirgpu_request_error(i).decode() for i in range(tok.n)]
        return out

    # -- device-resident split (bench / pipelining) -----------------------------------
    def stage_query(self, query) -> None:
        q = self._cts(query, 4, "query")
        self._check(self.lib.pirgpu_query_stage(self.db.handle, _ptr(q), q.shape[0]))

    def run_staged(self) -> None:
        self._check(self.lib.pirgpu_query_run(self.db.handle))

    def sync(self) -> None:
        self._check(self.lib.pirgpu_sync(self.db.handle))

    def device_synchronize(self) -> None:
        """hipDeviceSynchronize() through the library's own HIP runtime (every stream of the device)."""
        self._check(self.lib.pirgpu_device_synchronize(self.db.handle))

    def fetch_reply(self) -> np.ndarray:
        n = self.db.reply_ct_count()
        out = np.empty((n, 2, self.k, self.N), dtype=np.uint64)
        cnt = C.c_uint64(0)
        self._check(self.lib.pirgpu_query_fetch(self.db.handle, _ptr(out), n, C.byref(cnt)))
        return out[: cnt.value]

    # -- batch mode: several queries in flight (server.cpp:60-63 loop, overlapped on the GPU) ----
    def set_concurrency(self, n_workers: int) -> None:
        self._check(self.lib.pirgpu_set_concurrency(self.db.handle, n_workers))

    def stage_batch(self, queries) -> None:
        """queries: [count, nq, 2, k, N]"""
        q = self._cts(queries, 5, "queries")
        self._batch_count = q.shape[0]
        self._check(self.lib.pirgpu_batch_stage(self.db.handle, _ptr(q), q.shape[1], q.shape[0]))

    def run_batch(self) -> None:
        self._check(self.lib.pirgpu_batch_run(self.db.handle))

    def unstage_batch(self) -> None:
        """Forgets the staged queries -- and with them the references to key sets set_batch_keysets left, which keep
        those sets from being evicted for new clients."""
        self._check(self.lib.pirgpu_batch_unstage(self.db.handle))

    def fetch_batch(self) -> np.ndarray:
        n = self.db.reply_ct_count()
        out = np.empty((self._batch_count, n, 2, self.k, self.N), dtype=np.uint64)
        cnt = C.c_uint64(0)
        self._check(self.lib.pirgpu_batch_fetch(self.db.handle, _ptr(out), self._batch_count * n, C.byref(cnt)))
        return out

    def batch_expand(self, first: int, count: int, device_ptr: int) -> None:
        """Expansion + selector NTT of staged queries [first, first+count) into device memory."""
        self._check(self.lib.pirgpu_batch_expand(self.db.handle, first, count, C.c_void_p(device_ptr)))

    def batch_run_selectors(self, device_ptr: int, count: int) -> None:
        """PIRDatabase::multiply for `count` queries whose NTT-form selection vectors are on the device."""
        self._batch_count = count
        self._check(self.lib.pirgpu_batch_run_selectors(self.db.handle, C.c_void_p(device_ptr), count))

    def dim_sum(self) -> int:
        return self.params.dim_sum

    # -- packed selector exchange (row-sharded multi-GPU, d = 2) --------------------------
    def packed_selector_bytes(self) -> int:
        return int(self.lib.pirgpu_packed_selector_bytes(self.db.handle))

    def batch_expand_packed(self, first: int, count: int, packed_ptr: int, rows_ptr: int, row_cuts) -> None:
        cuts = (C.c_uint32 * len(row_cuts))(*row_cuts)
        self._check(self.lib.pirgpu_batch_expand_packed(self.db.handle, first, count, C.c_void_p(packed_ptr),
                                                        C.c_void_p(rows_ptr), cuts, len(row_cuts) - 1))

    # -- slot-sharded multi-GPU step (pirgpu_slots_*) -------------------------------
    def slots_packed_bytes(self, slots: int) -> int:
        """Bytes of `slots` NTT slots of one packed group of column selectors (0: the step does not apply)."""
        return int(self.lib.pirgpu_slots_packed_bytes(self.db.handle, int(slots)))

    def slots_expand_async(self, first: int, count: int, packed_ptr: int, sv_ptr: int, slot_cuts, after: int = 0,
                           then: int = 0) -> None:
        cuts = (C.c_uint32 * len(slot_cuts))(*slot_cuts)
        self._check(self.lib.pirgpu_slots_expand_async(self.db.handle, first, count, C.c_void_p(packed_ptr),
                                                       C.c_void_p(sv_ptr), cuts, len(slot_cuts) - 1,
                                                       C.c_void_p(after or None), C.c_void_p(then or None)))

    def slots_scan_async(self, packed_ptr: int, n_ranks: int, per_rank: int, rowsums_ptr: int, after: int = 0,
                         then: int = 0) -> None:
        self._check(self.lib.pirgpu_slots_scan_async(self.db.handle, C.c_void_p(packed_ptr), n_ranks, per_rank,
                                                     C.c_void_p(rowsums_ptr), C.c_void_p(after or None),
                                                     C.c_void_p(then or None)))

    def slots_finish_async(self, rowsums_ptr: int, count: int, sv_ptr: int, slot_cuts, replies_ptr: int, after: int = 0,
                           then: int = 0) -> None:
        cuts = (C.c_uint32 * len(slot_cuts))(*slot_cuts)
        self._check(self.lib.pirgpu_slots_finish_async(self.db.handle, C.c_void_p(rowsums_ptr), count, C.c_void_p(sv_ptr),
                                                       cuts, len(slot_cuts) - 1, C.c_void_p(replies_ptr),
                                                       C.c_void_p(after or None), C.c_void_p(then or None)))

    # -- the same without host synchronisation + device-side ordering (pipelined multi-GPU step) --------
    def stream_handle(self) -> int:
        """The library's main HIP stream (for torch.cuda.ExternalStream)."""
        return int(self.lib.pirgpu_stream_handle(self.db.handle) or 0)

    def join(self) -> None:
        """Main stream waits (on the device) for everything queued on the lanes / workers so far."""
        self._check(self.lib.pirgpu_join(self.db.handle))

    def join_stream(self, stream: int) -> None:
        """The caller's stream (a hipStream_t) waits for everything queued so far; the main stream does not."""
        self._check(self.lib.pirgpu_join_stream(self.db.handle, C.c_void_p(stream)))

    def batch_set_reply_buffer(self, device_ptr: int, capacity_cts: int) -> None:
        """Later batches write their replies into this device buffer (0 restores the context's own)."""
        self._check(self.lib.pirgpu_batch_set_reply_buffer(self.db.handle, C.c_void_p(device_ptr or None),
                                                           int(capacity_cts)))

    def fork(self) -> None:
        """Lanes / workers wait (on the device) for everything the main stream has been made to wait for."""
        self._check(self.lib.pirgpu_fork(self.db.handle))

    def batch_expand_packed_async(self, first: int, count: int, packed_ptr: int, rows_ptr: int, row_cuts) -> None:
        cuts = (C.c_uint32 * len(row_cuts))(*row_cuts)
        self._check(self.lib.pirgpu_batch_expand_packed_async(self.db.handle, first, count, C.c_void_p(packed_ptr),
                                                              C.c_void_p(rows_ptr), cuts, len(row_cuts) - 1))

    def batch_reply_copy_to_device_async(self, device_ptr: int) -> None:
        self._check(self.lib.pirgpu_batch_reply_copy_to_device_async(self.db.handle, C.c_void_p(device_ptr),
                                                                     self._batch_count * self.db.reply_ct_count()))

    def pack40_supported(self) -> bool:
        """Row selectors may cross GPUs in 5 bytes per residue (every data modulus below 2^40)."""
        return bool(self.lib.pirgpu_pack40_supported(self.db.handle))

    def pack40_async(self, words_ptr: int, packed_ptr: int, words: int, stream: int = 0) -> None:
        self._check(self.lib.pirgpu_pack40_device_async(self.db.handle, C.c_void_p(words_ptr), C.c_void_p(packed_ptr), words,
                                                        C.c_void_p(stream)))

    def unpack40_async(self, packed_ptr: int, words_ptr: int, words: int, stream: int = 0) -> None:
        self._check(self.lib.pirgpu_unpack40_device_async(self.db.handle, C.c_void_p(packed_ptr), C.c_void_p(words_ptr),
                                                          words, C.c_void_p(stream)))

    def reduce_fixup_device_async(self, device_ptr: int, n_cts: int, stream: int = 0) -> None:
        self._check(self.lib.pirgpu_reduce_fixup_device_async(self.db.handle, C.c_void_p(device_ptr), n_cts,
                                                              C.c_void_p(stream)))

    def batch_run_packed(self, packed_ptr: int, n_ranks: int, per_rank: int, rows_ptr: int) -> None:
        self._batch_count = n_ranks * per_rank
        self._check(self.lib.pirgpu_batch_run_packed(self.db.handle, C.c_void_p(packed_ptr), n_ranks, per_rank,
                                                     C.c_void_p(rows_ptr)))

    def zero_plaintexts(self) -> int:
        """Identically-zero plaintexts in this server's shard (SEAL's transparent-ciphertext condition)."""
        return int(self.lib.pirgpu_zero_plaintexts(self.db.handle))

    def set_remote_zero_plaintexts(self, count: int) -> None:
        """Zero plaintexts held by the OTHER shards of a row-sharded database (distributed.sync_zero_plaintexts)."""
        self._check(self.lib.pirgpu_set_remote_zero_plaintexts(self.db.handle, int(count)))

    def check_ready(self) -> None:
        """Raises what the next query would fail with (database not loaded / transparent) before anything runs."""
        self._check(self.lib.pirgpu_check_ready(self.db.handle))

    def shard_rows(self):
        """[begin, end) of dimension 0 held by this server's database."""
        p = self.db._cparams
        return (int(p.shard_begin), int(p.shard_end) if (p.shard_begin or p.shard_end) else self.params.dimensions[0])

    def batch_reply_copy_to_device(self, device_ptr: int) -> None:
        self._check(self.lib.pirgpu_batch_reply_copy_to_device(self.db.handle, C.c_void_p(device_ptr),
                                                               self._batch_count * self.db.reply_ct_count()))

    def reduce_fixup_device_n(self, device_ptr: int, n_cts: int) -> None:
        self._check(self.lib.pirgpu_reduce_fixup_device(self.db.handle, C.c_void_p(device_ptr), n_cts))

    def process_batch(self, queries, n_workers: Optional[int] = None) -> np.ndarray:
        """All queries of one request: [count, nq, 2, k, N] -> [count, reply_cts, 2, k, N]."""
        if n_workers is not None:
            self.set_concurrency(n_workers)
        self.stage_batch(queries)
        self.run_batch()
        return self.fetch_batch()

    def set_profiling(self, on: bool) -> None:
        self._check(self.lib.pirgpu_set_profiling(self.db.handle, 1 if on else 0))

    def last_timings(self) -> Dict[str, float]:
        ms = (C.c_float * 6)()
        runs = C.c_uint32(0)
        self._check(self.lib.pirgpu_last_timings(self.db.handle, ms, C.byref(runs)))
        names = ["expand_ms", "reserved", "scan_ms", "upper_ms", "final_ms", "total_ms"]
        out = {n: float(v) for n, v in zip(names, ms)}
        out["runs"] = int(runs.value)
        return out

    def reply_copy_to_device(self, device_ptr: int) -> None:
        """D2D copy of the last reply into caller-owned device memory (for the RCCL reduce)."""
        self._check(self.lib.pirgpu_reply_copy_to_device(self.db.handle, C.c_void_p(device_ptr),
                                                         self.db.reply_ct_count()))

    def reduce_fixup_device(self, device_ptr: int) -> None:
        """x <- x mod q_j over the summed partial replies, in place on the device."""
        self._check(self.lib.pirgpu_reduce_fixup_device(self.db.handle, C.c_void_p(device_ptr),
                                                        self.db.reply_ct_count()))

    def batch_scan_timings(self) -> Dict[str, float]:
        """Database-pass launches of the batches run since set_profiling(True): mean / min ms, how many, their shape."""
        mean, mn = C.c_float(0), C.c_float(0)
        n, wgs, nq = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        self._check(self.lib.pirgpu_batch_scan_timings(self.db.handle, C.byref(mean), C.byref(mn), C.byref(n), C.byref(wgs),
                                                       C.byref(nq)))
        return {"mean_ms": float(mean.value), "min_ms": float(mn.value), "launches": int(n.value),
                "workgroups": int(wgs.value), "queries": int(nq.value)}

    def ntt_mode(self) -> int:
        """0 integer / 1 fp64 / 2 wide fp64 butterflies (pirgpu_ntt_mode)."""
        return int(self.lib.pirgpu_ntt_mode(self.db.handle))

    def scan_bytes(self) -> int:
        return int(self.lib.pirgpu_scan_bytes(self.db.handle))

    def scan_info(self) -> dict:
        """How the database is scanned (pirgpu_scan_info): MFMA digit-sliced path or 64-bit MAC kernels."""
        info = (C.c_uint32 * 8)()
        self._check(self.lib.pirgpu_scan_info(self.db.handle, info))
        return {"mfma": bool(info[0]), "digits": info[1], "chunks": info[2], "ksteps": info[3],
                "queries_per_pass": info[4], "rows": info[5], "cols": info[6], "single_query_mfma": bool(info[7] & 1),
                "top_digit_nibble": bool(info[7] & 2)}

    # -- test-visible helpers (server.h:66-131) -----------------------------------------
    def substitute_power_x_inplace(self, ct: np.ndarray, power: int) -> np.ndarray:
        """server.cpp:67-76; returns the substituted ciphertext (ct itself is updated too)."""
        if not (isinstance(ct, np.ndarray) and ct.dtype == np.uint64 and ct.flags["C_CONTIGUOUS"]
                and ct.shape == (2, self.k, self.N)):
            raise PirGpuError(3, "ciphertext must be a contiguous uint64 array of shape [2, %d, %d]" % (self.k, self.N))
        self._check(self.lib.pirgpu_substitute_power_x(self.db.handle, _ptr(ct), power))
        return ct

    def multiply_inverse_power_of_x(self, ct, k: int) -> np.ndarray:
        """server.cpp:78-103"""
        ct = self._cts(ct, 3, "ciphertext")
        out = np.empty_like(ct)
        self._check(self.lib.pirgpu_multiply_inverse_power_of_x(self.db.handle, _ptr(ct), k, _ptr(out)))
        return out

    def oblivious_expansion(self, ct, num_items: int) -> np.ndarray:
        """server.cpp:105-146 (one ciphertext) or :148-171 (a list / 4-D array of ciphertexts)."""
        ct = _u64(ct)
        ct = self._cts(ct, 3 if ct.ndim == 3 else 4, "ciphertext(s)")
        out = np.empty((max(num_items, 1), 2, self.k, self.N), dtype=np.uint64)
        if ct.ndim == 3:
            self._check(self.lib.pirgpu_expand(self.db.handle, _ptr(ct), num_items, _ptr(out)))
        else:
            self._check(self.lib.pirgpu_expand_multi(self.db.handle, _ptr(ct), ct.shape[0], num_items, _ptr(out)))
        return out[:num_items]

    # -- NTT test hooks -------------------------------------------------------------------
    def ntt_forward(self, cts, key_level: bool = False) -> np.ndarray:
        a = _u64(cts).copy()
        count = a.shape[0]
        self._check(self.lib.pirgpu_ntt_forward(self.db.handle, _ptr(a), count, 1 if key_level else 0))
        return a

    def ntt_inverse(self, cts, key_level: bool = False) -> np.ndarray:
        a = _u64(cts).copy()
        count = a.shape[0]
        self._check(self.lib.pirgpu_ntt_inverse(self.db.handle, _ptr(a), count, 1 if key_level else 0))
        return a
