"""CPU oracle for the OpenMined/PIR server query path -- TEST INFRASTRUCTURE ONLY.

ctypes binding of ``oracle/pir_oracle.c`` (the C restatement of the reference's
``server.cpp`` / ``database.cpp`` / ``ct_reencoder.cpp`` / ``string_encoder.cpp``
/ ``utils.cpp`` plus the SEAL 3.5.6 primitives they call) and the parameter
shape math of ``parameters.cpp:56-107``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker.  The product
(``pir_amd``) never imports it.

Parity status: **parity unpinned** at ciphertext-bit level (SEAL is not
available in this image and the reference holds no ciphertext fixtures); pinned
on the reference tests' plaintext-level known answers (tests/test_oracle_*.py).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

OK, INVALID_ARGUMENT, INTERNAL = 0, 3, 13

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)


def build(native: bool = False) -> str:
    """Compile the C oracle (gcc). Returns the path of the shared object."""
    target = "native" if native else "all"
    name = "libpir_oracle_native.so" if native else "libpir_oracle.so"
    subprocess.run(["make", "-C", _HERE, target], check=True, capture_output=True)
    return os.path.join(_HERE, name)


def load(native: bool = False) -> C.CDLL:
    key = "native" if native else "portable"
    if key in _LIBS:
        return _LIBS[key]
    name = "libpir_oracle_native.so" if native else "libpir_oracle.so"
    path = os.path.join(_HERE, name)
    if os.environ.get("PIR_ORACLE_LIB"):      # e.g. the ASan/UBSan build (make -C oracle asan) under LD_PRELOAD
        path = os.environ["PIR_ORACLE_LIB"]
    src = os.path.join(_HERE, "pir_oracle.c")
    if not os.path.exists(path) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(path)):
        try:
            build(native)
        except Exception:
            if not os.path.exists(path):
                raise
    lib = C.CDLL(path)
    _declare(lib)
    _LIBS[key] = lib
    return lib


def _declare(lib):
    vp = C.c_void_p
    sigs = {
        "orc_create": (vp, [C.c_uint32, C.c_uint32, u64p, C.c_uint64]),
        "orc_destroy": (None, [vp]),
        "orc_psi": (C.c_uint64, [vp, C.c_uint32]),
        "orc_mulmod": (C.c_uint64, [C.c_uint64] * 3),
        "orc_powmod": (C.c_uint64, [C.c_uint64] * 3),
        "orc_invmod": (C.c_uint64, [C.c_uint64] * 2),
        "orc_is_prime": (C.c_int, [C.c_uint64]),
        "orc_minimal_primitive_root": (C.c_uint64, [C.c_uint64] * 2),
        "orc_ntt_fwd": (None, [vp, C.c_uint32, u64p]),
        "orc_ntt_inv": (None, [vp, C.c_uint32, u64p]),
        "orc_ct_ntt_fwd": (None, [vp, u64p]),
        "orc_ct_ntt_inv": (None, [vp, u64p]),
        "orc_dyadic_mul": (None, [vp, C.c_uint32, u64p, u64p, u64p]),
        "orc_poly_add": (None, [vp, C.c_uint32, u64p, u64p, u64p]),
        "orc_poly_sub": (None, [vp, C.c_uint32, u64p, u64p, u64p]),
        "orc_poly_neg": (None, [vp, C.c_uint32, u64p, u64p]),
        "orc_apply_galois_poly": (None, [vp, C.c_uint32, u64p, C.c_uint32, u64p]),
        "orc_negacyclic_shift_poly": (None, [vp, C.c_uint32, u64p, C.c_uint32, u64p]),
        "orc_divide_round_special": (None, [vp, u64p, u64p]),
        "orc_apply_galois_ct": (C.c_int, [vp, u64p, C.c_uint32, u64p]),
        "orc_multiply_inverse_power_of_x": (None, [vp, u64p, C.c_uint32, u64p]),
        "orc_ct_add_inplace": (None, [vp, u64p, u64p]),
        "orc_oblivious_expansion": (C.c_int, [vp, u64p, C.c_uint32, C.POINTER(u64p), u64p]),
        "orc_oblivious_expansion_multi": (C.c_int, [vp, u64p, C.c_uint32, C.c_uint64, C.POINTER(u64p), u64p]),
        "orc_plain_lift_ntt": (None, [vp, u64p, C.c_uint32, u64p]),
        "orc_multiply_plain_ntt": (None, [vp, u64p, u64p, u64p]),
        "orc_bits_per_coeff": (C.c_uint32, [C.c_uint64]),
        "orc_expansion_ratio": (C.c_uint32, [vp]),
        "orc_reencode": (None, [vp, u64p, u64p]),
        "orc_redecode": (None, [vp, u64p, u32p, u64p]),
        "orc_db_multiply": (C.c_int, [vp, u64p, C.c_uint64, u32p, C.c_uint32, u64p, u8p, C.c_uint64, u64p, u64p]),
        "orc_reply_ct_count": (C.c_uint64, [vp, C.c_uint32]),
        "orc_process_query": (C.c_int, [vp, u64p, C.c_uint64, u32p, C.c_uint32, u64p, C.c_uint32,
                                        C.POINTER(u64p), u64p, u64p]),
        "orc_items_per_plaintext": (C.c_uint64, [C.c_uint32, C.c_uint32, C.c_uint64]),
        "orc_max_bytes_per_plaintext": (C.c_uint64, [C.c_uint32, C.c_uint32]),
        "orc_string_encode": (C.c_int, [u8p, C.c_uint64, C.c_uint32, C.c_uint32, u64p, u32p]),
        "orc_string_decode": (C.c_int, [u64p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, u8p]),
        "orc_db_encode": (C.c_int, [vp, u8p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, u64p, C.c_uint64]),
        "orc_calculate_dimensions": (None, [C.c_uint32, C.c_uint32, u32p]),
        "orc_calculate_indices": (None, [C.c_uint32, C.c_uint32, u32p, C.c_uint32, u32p]),
        "orc_calculate_item_offset": (C.c_uint64, [C.c_uint32, C.c_uint32, C.c_uint32]),
        "orc_ceil_log2": (C.c_uint32, [C.c_uint32]),
        "orc_log2": (C.c_uint32, [C.c_uint32]),
        "orc_next_power_two": (C.c_uint64, [C.c_uint64]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(u64p)


# ----------------------------------------------------------------------------
# SEAL parameter tables (values verified prime, == 1 mod 2N)
# ----------------------------------------------------------------------------

BFV_DEFAULT = {
    2048: [0x3FFFFFFF000001],
    4096: [0xFFFFEE001, 0xFFFFC4001, 0x1FFFFE0001],
    8192: [0x7FFFFFD8001, 0x7FFFFFC8001, 0xFFFFFFFC001, 0xFFFFFF6C001, 0xFFFFFEBC001],
    16384: [0xFFFFFFFD8001, 0xFFFFFFFA0001, 0xFFFFFFF00001, 0x1FFFFFFF68001, 0x1FFFFFFF50001,
            0x1FFFFFFEE8001, 0x1FFFFFFEA0001, 0x1FFFFFFE88001, 0x1FFFFFFE48001],
}


def is_prime(n: int) -> bool:
    return bool(load().orc_is_prime(n))


def coeff_modulus_create(N: int, bit_sizes: Sequence[int]) -> List[int]:
    """SEAL CoeffModulus::Create(N, bit_sizes): per bit size the largest primes
    == 1 (mod 2N) below 2^bits, found descending, handed out smallest-first."""
    need = {}
    for b in bit_sizes:
        need[b] = need.get(b, 0) + 1
    table = {}
    for b, cnt in need.items():
        found, v, lo = [], (1 << b) - 2 * N + 1, 1 << (b - 1)
        while cnt and v > lo:
            if is_prime(v):
                found.append(v)
                cnt -= 1
            v -= 2 * N
        if cnt:
            raise ValueError("not enough primes")
        table[b] = found
    return [table[b].pop() for b in bit_sizes]


def plain_modulus_batching(N: int, bits: int) -> int:
    """SEAL PlainModulus::Batching(N, bits)."""
    return coeff_modulus_create(N, [bits])[0]


# ----------------------------------------------------------------------------
# PIRParameters (payload.proto:45-69) + CreatePIRParameters (parameters.cpp:56-107)
# ----------------------------------------------------------------------------

@dataclass
class PirParams:
    N: int
    moduli: List[int]            # k data primes followed by the special prime
    t: int
    num_items: int
    num_pt: int
    dimensions: List[int]
    bytes_per_item: int
    items_per_plaintext: int
    bits_per_coeff: int = 0      # 0 = default floor(log2 t)
    use_ciphertext_multiplication: bool = False

    @property
    def k(self) -> int:
        return len(self.moduli) - 1

    @property
    def data_moduli(self) -> List[int]:
        return self.moduli[:-1]

    @property
    def special(self) -> int:
        return self.moduli[-1]

    @property
    def dim_sum(self) -> int:
        return sum(self.dimensions)

    @property
    def eff_bits_per_coeff(self) -> int:
        return self.bits_per_coeff if self.bits_per_coeff > 0 else bits_per_coeff(self.t)


def bits_per_coeff(t: int) -> int:
    return int(load().orc_bits_per_coeff(t))


def calculate_dimensions(db_size: int, nd: int) -> List[int]:
    out = (C.c_uint32 * nd)()
    load().orc_calculate_dimensions(db_size, nd, out)
    return list(out)


def create_pir_parameters(dbsize: int, bytes_per_item: int = 0, dimensions: int = 1, N: int = 4096,
                          plain_bits: int = 20, moduli: Optional[Sequence[int]] = None, t: Optional[int] = None,
                          use_ciphertext_multiplication: bool = False, bits_per_coeff_: int = 0) -> PirParams:
    """CreatePIRParameters (parameters.cpp:56-107) over GenerateEncryptionParams
    (parameters.cpp:33-54: BFVDefault coefficient modulus, Batching plain modulus)."""
    lib = load()
    if moduli is None:
        moduli = BFV_DEFAULT[N]
    if t is None:
        t = plain_modulus_batching(N, plain_bits)
    bpc_default = bits_per_coeff(t)
    bpc = bpc_default
    if bits_per_coeff_ > 0:
        if bits_per_coeff_ > bpc_default:
            raise ValueError("Bits per coefficient greater than max")
        bpc = bits_per_coeff_
    if bytes_per_item > 0:
        ipp = int(lib.orc_items_per_plaintext(N, bpc, bytes_per_item))
        if ipp <= 0:
            raise ValueError("Cannot fit an item within one plaintext")
        num_pt = dbsize // ipp
        while dbsize > num_pt * ipp:
            num_pt += 1
        bpi = bytes_per_item
    else:
        bpi = int(lib.orc_max_bytes_per_plaintext(N, bpc))
        ipp = 1
        num_pt = dbsize
    return PirParams(N=N, moduli=list(moduli), t=t, num_items=dbsize, num_pt=num_pt,
                     dimensions=calculate_dimensions(num_pt, dimensions), bytes_per_item=bpi,
                     items_per_plaintext=ipp, bits_per_coeff=bits_per_coeff_,
                     use_ciphertext_multiplication=use_ciphertext_multiplication)


# ----------------------------------------------------------------------------
# Oracle context
# ----------------------------------------------------------------------------

class Oracle:
    """Thin numpy front-end of the C oracle for one (N, moduli, t)."""

    def __init__(self, N: int, moduli: Sequence[int], t: int, native: bool = False):
        self.lib = load(native)
        self.N, self.t = N, t
        self.moduli = list(moduli)
        self.k = len(moduli) - 1
        arr = (C.c_uint64 * (self.k + 1))(*moduli)
        self.ctx = self.lib.orc_create(N, self.k, arr, t)
        if not self.ctx:
            raise ValueError("invalid encryption parameters")
        self.logN = N.bit_length() - 1
        self.ct_words = 2 * self.k * N

    @classmethod
    def from_params(cls, p: PirParams, native: bool = False) -> "Oracle":
        return cls(p.N, p.moduli, p.t, native)

    def __del__(self):
        try:
            self.lib.orc_destroy(self.ctx)
        except Exception:
            pass

    # -- primitives ---------------------------------------------------------
    def psi(self, mi):
        return int(self.lib.orc_psi(self.ctx, mi))

    def ntt_fwd(self, mi, poly):
        a = np.ascontiguousarray(poly, dtype=np.uint64).copy()
        self.lib.orc_ntt_fwd(self.ctx, mi, _p(a))
        return a

    def ntt_inv(self, mi, poly):
        a = np.ascontiguousarray(poly, dtype=np.uint64).copy()
        self.lib.orc_ntt_inv(self.ctx, mi, _p(a))
        return a

    def ct_ntt_fwd(self, ct):
        a = np.ascontiguousarray(ct, dtype=np.uint64).copy()
        self.lib.orc_ct_ntt_fwd(self.ctx, _p(a))
        return a

    def ct_ntt_inv(self, ct):
        a = np.ascontiguousarray(ct, dtype=np.uint64).copy()
        self.lib.orc_ct_ntt_inv(self.ctx, _p(a))
        return a

    def dyadic_mul(self, mi, a, b):
        out = np.empty(self.N, dtype=np.uint64)
        self.lib.orc_dyadic_mul(self.ctx, mi, _p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)), _p(out))
        return out

    def poly_add(self, mi, a, b):
        out = np.empty(self.N, dtype=np.uint64)
        self.lib.orc_poly_add(self.ctx, mi, _p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)), _p(out))
        return out

    def poly_sub(self, mi, a, b):
        out = np.empty(self.N, dtype=np.uint64)
        self.lib.orc_poly_sub(self.ctx, mi, _p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)), _p(out))
        return out

    def poly_neg(self, mi, a):
        out = np.empty(self.N, dtype=np.uint64)
        self.lib.orc_poly_neg(self.ctx, mi, _p(np.ascontiguousarray(a)), _p(out))
        return out

    def apply_galois_poly(self, mi, a, g):
        out = np.empty(self.N, dtype=np.uint64)
        self.lib.orc_apply_galois_poly(self.ctx, mi, _p(np.ascontiguousarray(a)), g, _p(out))
        return out

    def negacyclic_shift_poly(self, mi, a, shift):
        out = np.empty(self.N, dtype=np.uint64)
        self.lib.orc_negacyclic_shift_poly(self.ctx, mi, _p(np.ascontiguousarray(a)), shift, _p(out))
        return out

    def divide_round_special(self, x):
        """x: [(k+1), N] coefficient form over all key-level moduli -> [k, N]."""
        x = np.ascontiguousarray(x, dtype=np.uint64)
        out = np.empty((self.k, self.N), dtype=np.uint64)
        self.lib.orc_divide_round_special(self.ctx, _p(x), _p(out))
        return out

    # -- ciphertext-level ops (reference server.cpp) ------------------------
    def new_ct(self, n=1):
        return np.zeros((n, 2, self.k, self.N), dtype=np.uint64)

    def apply_galois_ct(self, ct, g, key):
        a = np.ascontiguousarray(ct, dtype=np.uint64).copy()
        rc = self.lib.orc_apply_galois_ct(self.ctx, _p(a), g, _p(key) if key is not None else None)
        return rc, a

    def multiply_inverse_power_of_x(self, ct, kpow):
        out = np.empty_like(ct)
        self.lib.orc_multiply_inverse_power_of_x(self.ctx, _p(np.ascontiguousarray(ct)), kpow, _p(out))
        return out

    def _keys_arg(self, galois_keys):
        """galois_keys: dict {galois_elt: ndarray[k,2,k+1,N]} -> u64p[logN] indexed by level j."""
        arr = (u64p * self.logN)()
        self._keepalive = []
        for j in range(self.logN):
            key = galois_keys.get((self.N >> j) + 1) if galois_keys else None
            if key is not None:
                key = np.ascontiguousarray(key, dtype=np.uint64)
                self._keepalive.append(key)
                arr[j] = _p(key)
            else:
                arr[j] = None
        return arr

    def oblivious_expansion(self, ct, num_items, galois_keys):
        out = np.zeros((max(num_items, 1), 2, self.k, self.N), dtype=np.uint64)
        rc = self.lib.orc_oblivious_expansion(self.ctx, _p(np.ascontiguousarray(ct, dtype=np.uint64)), num_items,
                                              self._keys_arg(galois_keys), _p(out))
        return rc, out[:num_items]

    def oblivious_expansion_multi(self, cts, total_items, galois_keys):
        cts = np.ascontiguousarray(cts, dtype=np.uint64)
        out = np.zeros((max(total_items, 1), 2, self.k, self.N), dtype=np.uint64)
        rc = self.lib.orc_oblivious_expansion_multi(self.ctx, _p(cts), cts.shape[0], total_items,
                                                    self._keys_arg(galois_keys), _p(out))
        return rc, out[:total_items]

    # -- plaintext / database ------------------------------------------------
    def plain_lift_ntt(self, coeffs):
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
        out = np.empty((self.k, self.N), dtype=np.uint64)
        self.lib.orc_plain_lift_ntt(self.ctx, _p(coeffs), coeffs.shape[0], _p(out))
        return out

    def multiply_plain_ntt(self, ct_ntt, pt_ntt):
        out = np.empty((2, self.k, self.N), dtype=np.uint64)
        self.lib.orc_multiply_plain_ntt(self.ctx, _p(np.ascontiguousarray(ct_ntt)), _p(np.ascontiguousarray(pt_ntt)),
                                        _p(out))
        return out

    def expansion_ratio(self):
        return int(self.lib.orc_expansion_ratio(self.ctx))

    def reencode(self, ct):
        out = np.empty((2 * self.expansion_ratio(), self.N), dtype=np.uint64)
        self.lib.orc_reencode(self.ctx, _p(np.ascontiguousarray(ct, dtype=np.uint64)), _p(out))
        return out

    def redecode(self, pts):
        pts = np.ascontiguousarray(pts, dtype=np.uint64)
        out = np.empty((2, self.k, self.N), dtype=np.uint64)
        self.lib.orc_redecode(self.ctx, _p(pts), None, _p(out))
        return out

    def reply_ct_count(self, nd):
        return int(self.lib.orc_reply_ct_count(self.ctx, nd))

    def db_encode(self, items: bytes, num_items, bytes_per_item, items_per_pt, bits, num_pt):
        buf = np.frombuffer(items, dtype=np.uint8)
        out = np.empty((num_pt, self.k, self.N), dtype=np.uint64)
        rc = self.lib.orc_db_encode(self.ctx, buf.ctypes.data_as(u8p), num_items, bytes_per_item, items_per_pt, bits,
                                    _p(out), num_pt)
        return rc, out

    def db_from_coeffs(self, coeff_rows):
        """list of coefficient arrays (< t) -> [P, k, N] NTT plaintexts (database.cpp:74,104)."""
        out = np.empty((len(coeff_rows), self.k, self.N), dtype=np.uint64)
        for i, row in enumerate(coeff_rows):
            out[i] = self.plain_lift_ntt(np.asarray(row, dtype=np.uint64))
        return out

    def db_multiply(self, db_ntt, dims, sv, sv_is_ntt=None):
        """PIRDatabase::multiply (database.cpp:290-316). sv is mutated like the reference's."""
        db_ntt = np.ascontiguousarray(db_ntt, dtype=np.uint64)
        assert sv.dtype == np.uint64 and sv.flags["C_CONTIGUOUS"]
        nd = len(dims)
        d = (C.c_uint32 * nd)(*dims)
        if sv_is_ntt is None:
            sv_is_ntt = np.zeros(sv.shape[0], dtype=np.uint8)
        out = np.zeros((self.reply_ct_count(nd), 2, self.k, self.N), dtype=np.uint64)
        cnt = C.c_uint64(0)
        rc = self.lib.orc_db_multiply(self.ctx, _p(db_ntt), db_ntt.shape[0], d, nd, _p(sv),
                                      sv_is_ntt.ctypes.data_as(u8p), sv.shape[0], _p(out), C.byref(cnt))
        return rc, out[:cnt.value]

    def process_query(self, db_ntt, dims, query_cts, galois_keys):
        """processQuery (server.cpp:173-195) on residue arrays."""
        db_ntt = np.ascontiguousarray(db_ntt, dtype=np.uint64)
        query_cts = np.ascontiguousarray(query_cts, dtype=np.uint64)
        nd = len(dims)
        d = (C.c_uint32 * nd)(*dims)
        out = np.zeros((self.reply_ct_count(nd), 2, self.k, self.N), dtype=np.uint64)
        cnt = C.c_uint64(0)
        rc = self.lib.orc_process_query(self.ctx, _p(db_ntt), db_ntt.shape[0], d, nd, _p(query_cts),
                                        query_cts.shape[0], self._keys_arg(galois_keys), _p(out), C.byref(cnt))
        return rc, out[:cnt.value]


# ----------------------------------------------------------------------------
# string encoder / index math / utils front-ends
# ----------------------------------------------------------------------------

def string_encode(data: bytes, bits: int, N: int):
    lib = load()
    buf = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(0, dtype=np.uint8)
    coeffs = np.zeros(N, dtype=np.uint64)
    nc = C.c_uint32(0)
    rc = lib.orc_string_encode(buf.ctypes.data_as(u8p), len(data), bits, N, _p(coeffs), C.byref(nc))
    return rc, coeffs[:nc.value].copy()


def string_decode(coeffs, bits: int, length: int, byte_offset: int = 0):
    lib = load()
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
    out = np.zeros(max(length, 1), dtype=np.uint8)
    rc = lib.orc_string_decode(_p(coeffs), coeffs.shape[0], bits, length, byte_offset, out.ctypes.data_as(u8p))
    return rc, out[:length].tobytes()


def calculate_indices(index, items_per_pt, dims):
    nd = len(dims)
    out = (C.c_uint32 * nd)()
    load().orc_calculate_indices(index, items_per_pt, (C.c_uint32 * nd)(*dims), nd, out)
    return list(out)


def calculate_item_offset(index, items_per_pt, bytes_per_item):
    return int(load().orc_calculate_item_offset(index, items_per_pt, bytes_per_item))


def ceil_log2(v):
    return int(load().orc_ceil_log2(v))


def log2(v):
    return int(load().orc_log2(v))


def next_power_two(n):
    return int(load().orc_next_power_two(n))


def generate_galois_elts(N: int) -> List[int]:
    """reference utils.cpp:7-14"""
    return [(N >> i) + 1 for i in range(ceil_log2(N))]
