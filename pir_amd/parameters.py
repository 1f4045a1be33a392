"""Host-side mirror of the reference's parameter layer (stays on the CPU per north_star).

* ``GenerateEncryptionParams``  -- reference parameters.cpp:26-54
* ``CreatePIRParameters``       -- reference parameters.cpp:56-107
* ``PIRDatabase::calculate_dimensions / calculate_indices / calculate_item_offset``
                                -- reference database.cpp:318-342
* ``generate_galois_elts / next_power_two / ceil_log2`` -- reference utils.cpp, utils.h

Product code: does not import ``oracle``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Sequence

# SEAL 3.5.6 CoeffModulus::BFVDefault tables; last prime = key-switching prime.
BFV_DEFAULT = {
    4096: [0xFFFFEE001, 0xFFFFC4001, 0x1FFFFE0001],
    8192: [0x7FFFFFD8001, 0x7FFFFFC8001, 0xFFFFFFFC001, 0xFFFFFF6C001, 0xFFFFFEBC001],
    16384: [0xFFFFFFFD8001, 0xFFFFFFFA0001, 0xFFFFFFF00001, 0x1FFFFFFF68001, 0x1FFFFFFF50001,
            0x1FFFFFFEE8001, 0x1FFFFFFEA0001, 0x1FFFFFFE88001, 0x1FFFFFFE48001],
}


def _is_prime(n: int) -> bool:
    if n < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % p == 0:
            return n == p
    d, r = n - 1, 0
    while d % 2 == 0:
        d //= 2
        r += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(r - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def coeff_modulus_create(N: int, bit_sizes: Sequence[int]) -> List[int]:
    """SEAL CoeffModulus::Create: per bit size the largest primes == 1 (mod 2N) below
    2^bits (found descending), handed out smallest-first."""
    need = {}
    for b in bit_sizes:
        need[b] = need.get(b, 0) + 1
    table = {}
    for b, cnt in need.items():
        found, v, lo = [], (1 << b) - 2 * N + 1, 1 << (b - 1)
        while cnt and v > lo:
            if _is_prime(v):
                found.append(v)
                cnt -= 1
            v -= 2 * N
        if cnt:
            raise ValueError("failed to find enough qualifying primes")
        table[b] = found
    return [table[b].pop() for b in bit_sizes]


def plain_modulus_batching(N: int, bits: int) -> int:
    """SEAL PlainModulus::Batching."""
    return coeff_modulus_create(N, [bits])[0]


def next_power_two(n: int) -> int:  # utils.h:29-37
    if n == 0:
        return 1
    return 1 << (n - 1).bit_length()


def ceil_log2(v: int) -> int:  # utils.cpp:30-44
    return 0 if v <= 1 else (v - 1).bit_length()


def generate_galois_elts(N: int) -> List[int]:  # utils.cpp:7-14
    return [(N >> i) + 1 for i in range(ceil_log2(N))]


def bits_per_coeff(t: int) -> int:
    """floor(log2 t) via double log2, as string_encoder.cpp:85 / ct_reencoder.cpp:32."""
    return int(math.log2(float(t)))


def calculate_dimensions(db_size: int, num_dimensions: int) -> List[int]:  # database.cpp:334-342
    out = []
    for i in range(num_dimensions, 0, -1):
        out.append(int(math.ceil(math.pow(float(db_size), 1.0 / i))))
        db_size = int(math.ceil(float(db_size) / out[-1]))
    return out


@dataclass
class EncryptionParams:
    """What GenerateEncryptionParams returns (BFV; SEAL EncryptionParameters)."""
    poly_modulus_degree: int
    coeff_modulus: List[int]      # data primes followed by the special prime
    plain_modulus: int


def generate_encryption_params(poly_modulus_degree: int = 4096, plain_mod_bit_size: int = 20,
                               coeff_modulus: Optional[Sequence[int]] = None,
                               plain_modulus: Optional[int] = None) -> EncryptionParams:
    if coeff_modulus is None:
        coeff_modulus = BFV_DEFAULT[poly_modulus_degree]
    if plain_modulus is None:
        plain_modulus = plain_modulus_batching(poly_modulus_degree, plain_mod_bit_size)
    return EncryptionParams(poly_modulus_degree, list(coeff_modulus), plain_modulus)


@dataclass
class PIRParameters:
    """pir/proto/payload.proto:45-69 with encryption_parameters kept structured."""
    num_items: int
    num_pt: int
    dimensions: List[int]
    encryption_parameters: EncryptionParams
    bytes_per_item: int
    items_per_plaintext: int
    bits_per_coeff: int = 0
    use_ciphertext_multiplication: bool = False

    @property
    def dim_sum(self) -> int:
        return sum(self.dimensions)

    def calculate_indices(self, index: int) -> List[int]:  # database.cpp:318-326
        pt_index = index // self.items_per_plaintext
        out = [0] * len(self.dimensions)
        for i in range(len(out) - 1, -1, -1):
            out[i] = pt_index % self.dimensions[i]
            pt_index //= self.dimensions[i]
        return out

    def calculate_item_offset(self, index: int) -> int:  # database.cpp:328-332
        pt_index = index // self.items_per_plaintext
        return (index - pt_index * self.items_per_plaintext) * self.bytes_per_item


def create_pir_parameters(dbsize: int, bytes_per_item: int = 0, dimensions: int = 1,
                          enc: Optional[EncryptionParams] = None, use_ciphertext_multiplication: bool = False,
                          bits_per_coeff_: int = 0) -> PIRParameters:
    """CreatePIRParameters (parameters.cpp:56-107); raises ValueError where it returns InvalidArgument."""
    if enc is None:
        enc = generate_encryption_params()
    N, t = enc.poly_modulus_degree, enc.plain_modulus
    bpc = bits_per_coeff(t)
    if bits_per_coeff_ > 0:
        if bits_per_coeff_ > bpc:
            raise ValueError("Bits per coefficient greater than max")
        bpc = bits_per_coeff_
    if bytes_per_item > 0:
        ipp = N * bpc // bytes_per_item // 8          # string_encoder.cpp:25-27
        if ipp <= 0:
            raise ValueError("Cannot fit an item within one plaintext")
        num_pt = dbsize // ipp
        while dbsize > num_pt * ipp:
            num_pt += 1
        bpi = bytes_per_item
    else:
        bpi = N * bpc // 8                            # string_encoder.cpp:29-31
        ipp = 1
        num_pt = dbsize
    return PIRParameters(num_items=dbsize, num_pt=num_pt, dimensions=calculate_dimensions(num_pt, dimensions),
                         encryption_parameters=enc, bytes_per_item=bpi, items_per_plaintext=ipp,
                         bits_per_coeff=bits_per_coeff_, use_ciphertext_multiplication=use_ciphertext_multiplication)
