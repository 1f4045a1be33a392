"""PIRClient -- Python mirror of the reference client (pir/cpp/client.h:34-97) over libpirclient.so.

CPU only (the client stays on the CPU in the reference too); all arithmetic -- key generation,
encryption, decryption, recursive reply decode -- is in ``csrc/client.cpp`` behind the C ABI of
``include/pirclient.h``.  This module marshals numpy buffers and bytes.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import capi
from .parameters import PIRParameters, generate_galois_elts
from .server import PirGpuError


def _u64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(capi.u64p)


class PIRClient:
    """reference client.h:34-97.  ``seed`` makes key generation and encryption deterministic (tests)."""

    def __init__(self, params: PIRParameters, seed: Optional[bytes] = None):
        self.params = params
        enc = params.encryption_parameters
        self.N = enc.poly_modulus_degree
        self.k = len(enc.coeff_modulus) - 1
        self.lib = capi.load_client()
        p = capi.make_params(params)
        h = C.c_void_p()
        if seed is None:
            rc = self.lib.pirclient_create(C.byref(p), None, 0, C.byref(h))
        else:
            sb = (C.c_uint8 * len(seed)).from_buffer_copy(seed) if seed else (C.c_uint8 * 1)()
            rc = self.lib.pirclient_create(C.byref(p), sb, len(seed), C.byref(h))
        if rc != 0:
            raise PirGpuError(rc, self.lib.pirclient_create_error().decode())
        self._h = h

    @classmethod
    def Create(cls, params: PIRParameters, seed: Optional[bytes] = None) -> "PIRClient":
        """client.cpp:61-67."""
        return cls(params, seed)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.pirclient_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise PirGpuError(rc, self.lib.pirclient_last_error(self._h).decode())

    def set_seeded_keys(self, enabled: bool = True) -> None:
        """Key fields of CreateRequest: seed-compressed Serializable<> objects (default, what the reference client
        sends, client.cpp:47-54) or fully expanded ones."""
        self._check(self.lib.pirclient_set_seeded_keys(self._h, 1 if enabled else 0))

    # -- reference interface -------------------------------------------------------
    def CreateRequest(self, indexes: Sequence[int]) -> bytes:
        """client.cpp:80-90 -> serialized pir.Request."""
        idx = _u64(list(indexes))
        out, n = C.c_void_p(), C.c_size_t()
        self._check(self.lib.pirclient_create_request(self._h, _ptr(idx), idx.shape[0], C.byref(out), C.byref(n)))
        try:
            return C.string_at(out, n.value)
        finally:
            self.lib.pirclient_free(out)

    def SaveRequest(self, queries: Sequence[np.ndarray]) -> bytes:
        """serialization.cpp:44-73 with this client's keys: a serialized pir.Request around caller-supplied query
        ciphertexts (each [query_ct_count, 2, k, N], e.g. from create_query_for) -- the same ciphertexts can then go
        through the residue-level and the wire-level server entry points."""
        q = _u64(np.stack([_u64(x).reshape(self.query_ct_count, 2, self.k, self.N) for x in queries])) if len(queries) \
            else np.zeros((0,), dtype=np.uint64)
        out, n = C.c_void_p(), C.c_size_t()
        self._check(self.lib.pirclient_save_request(self._h, _ptr(q), len(queries), C.byref(out), C.byref(n)))
        try:
            return C.string_at(out, n.value)
        finally:
            self.lib.pirclient_free(out)

    def LoadResponse(self, response: bytes, max_replies: int = 4096) -> np.ndarray:
        """serialization.cpp:32-42 over every Response.reply -> [n_replies, reply_ct_count, 2, k, N] uint64."""
        buf = (C.c_uint8 * max(1, len(response))).from_buffer_copy(response or b"\0")
        # a reply is at least one ciphertext object: bound the output by the response's size
        per = self.reply_ct_count * 2 * self.k * self.N
        cap = max(1, min(max_replies, len(response) // (per * 8) + 1))
        out = np.empty((cap, self.reply_ct_count, 2, self.k, self.N), dtype=np.uint64)
        n = C.c_size_t()
        self._check(self.lib.pirclient_load_response(self._h, buf, len(response), _ptr(out), cap, C.byref(n)))
        return out[:n.value]

    def ProcessResponse(self, indexes: Sequence[int], response: bytes) -> List[bytes]:
        """client.cpp:160-185: serialized pir.Response -> one item per index."""
        idx = _u64(list(indexes))
        item = self.params.bytes_per_item
        out = np.zeros(max(1, idx.shape[0] * item), dtype=np.uint8)
        buf = (C.c_uint8 * max(1, len(response))).from_buffer_copy(response or b"\0")
        self._check(self.lib.pirclient_process_response(self._h, _ptr(idx), idx.shape[0], buf, len(response),
                                                        out.ctypes.data_as(capi.u8p), idx.shape[0] * item))
        return [out[i * item:(i + 1) * item].tobytes() for i in range(idx.shape[0])]

    def ProcessResponseInteger(self, response: bytes, max_replies: int = 1024) -> List[int]:
        """client.cpp:146-158."""
        out = np.zeros(max_replies, dtype=np.int64)
        n = C.c_size_t()
        buf = (C.c_uint8 * max(1, len(response))).from_buffer_copy(response or b"\0")
        self._check(self.lib.pirclient_process_response_integer(self._h, buf, len(response),
                                                                out.ctypes.data_as(capi.i64p), max_replies,
                                                                C.byref(n)))
        return [int(v) for v in out[:n.value]]

    # -- residue level ---------------------------------------------------------------
    @property
    def query_ct_count(self) -> int:
        return int(self.lib.pirclient_query_ct_count(self._h))

    @property
    def reply_ct_count(self) -> int:
        return int(self.lib.pirclient_reply_ct_count(self._h))

    def create_query_for(self, index: int) -> np.ndarray:
        """client.cpp:92-144 -> [n, 2, k, N] uint64."""
        n = self.query_ct_count
        out = np.empty((n, 2, self.k, self.N), dtype=np.uint64)
        got = C.c_uint32()
        self._check(self.lib.pirclient_create_query(self._h, int(index), _ptr(out), n, C.byref(got)))
        return out[:got.value]

    def galois_key(self, elt: int) -> np.ndarray:
        out = np.empty((self.k, 2, self.k + 1, self.N), dtype=np.uint64)
        self._check(self.lib.pirclient_galois_key(self._h, int(elt), _ptr(out)))
        return out

    def galois_keys(self) -> Dict[int, np.ndarray]:
        """The keys of initialize() (client.cpp:47) as residues, for PIRServer.set_galois_keys."""
        return {g: self.galois_key(g) for g in generate_galois_elts(self.N)}

    def process_reply(self, reply: np.ndarray) -> np.ndarray:
        """client.cpp:187-255 -> plaintext coefficients [N]."""
        r = _u64(reply)
        out = np.empty(self.N, dtype=np.uint64)
        self._check(self.lib.pirclient_process_reply(self._h, _ptr(r), r.shape[0], _ptr(out)))
        return out

    # -- the SEAL objects the reference tests reach through friend access --------------
    def encrypt(self, plaintext) -> np.ndarray:
        pt = _u64(plaintext)
        out = np.empty((2, self.k, self.N), dtype=np.uint64)
        self._check(self.lib.pirclient_encrypt(self._h, _ptr(pt), pt.shape[0], _ptr(out)))
        return out

    def decrypt(self, ct) -> np.ndarray:
        c = _u64(ct)
        out = np.empty(self.N, dtype=np.uint64)
        self._check(self.lib.pirclient_decrypt(self._h, _ptr(c), _ptr(out)))
        return out

    def noise_budget(self, ct) -> int:
        c = _u64(ct)
        bits = C.c_int()
        self._check(self.lib.pirclient_noise_budget(self._h, _ptr(c), C.byref(bits)))
        return bits.value

    def reencode(self, ct) -> np.ndarray:
        """CiphertextReencoder::Encode (ct_reencoder.cpp:40-73) -> [2 * ratio, N]."""
        c = _u64(ct)
        enc = self.params.encryption_parameters
        b = int(math.log2(enc.plain_modulus))
        cap = 2 * sum(math.ceil(math.log2(q) / b) for q in enc.coeff_modulus[:-1])
        out = np.empty((cap, self.N), dtype=np.uint64)
        n = C.c_uint32()
        self._check(self.lib.pirclient_reencode(self._h, _ptr(c), _ptr(out), cap, C.byref(n)))
        return out[:n.value].copy()

    def string_decode(self, plaintext, length: int, byte_offset: int = 0) -> bytes:
        pt = _u64(plaintext)
        out = np.zeros(max(1, length), dtype=np.uint8)
        self._check(self.lib.pirclient_string_decode(self._h, _ptr(pt), length, byte_offset,
                                                     out.ctypes.data_as(capi.u8p)))
        return out[:length].tobytes()
