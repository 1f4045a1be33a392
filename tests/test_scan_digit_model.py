"""The arithmetic behind the scan's operand format (pir_amd/csrc/scan_mfma.hip), restated in numpy and checked on the CPU:
balanced base-256 digits with the TOP digit stored as a nibble.  A residue x of a modulus q <= 2^(8 (L-1) + 4) is centred
asymmetrically -- v = x for x <= vmax, else x - q, vmax = 7 * 256^(L-1) + 127 (256^(L-1) - 1) / 255 -- so that its balanced
low digits leave a top digit in [-8, 7]; a row of 16 top digits packs into 8 bytes (byte i of the first word = columns i
and i + 4, of the second word columns 8 + i and 12 + i) and unpacks with two shifts, two masks and one multiply per word.
The GPU tests check the kernels themselves (every geometry in both forms, boundary residues against the 64-bit scan);
this file pins the claims the kernel comments make."""
import numpy as np
import pytest

import oracle


def vmax(L):
    p = 256 ** (L - 1)
    return 7 * p + 127 * ((p - 1) // 255)


def to_digits_top4(x, q, L):
    v = np.where(x > vmax(L), x.astype(object) - q, x.astype(object))
    digits = []
    for _ in range(L - 1):
        d = ((v + 128) % 256) - 128            # balanced low digit in [-128, 127]
        digits.append(d)
        v = (v - d) // 256
    digits.append(v)
    return digits


def sx4(v):
    return (v | (((v >> 3) & 0x01010101) * 0xF0)) & 0xFFFFFFFF


def expand_top4(w0, w1):
    words = [sx4(w0 & 0x0F0F0F0F), sx4((w0 >> 4) & 0x0F0F0F0F), sx4(w1 & 0x0F0F0F0F), sx4((w1 >> 4) & 0x0F0F0F0F)]
    out = []
    for w in words:
        for i in range(4):
            b = (w >> (8 * i)) & 0xFF
            out.append(b - 256 if b >= 128 else b)
    return out


def pack_top4(row):
    w = [0, 0]
    for h in range(2):
        for i in range(4):
            w[h] |= ((row[8 * h + i] & 0xF) | ((row[8 * h + 4 + i] & 0xF) << 4)) << (8 * i)
    return w


@pytest.mark.parametrize("L,bits,q", [(5, 36, None), (5, 36, 2 ** 36 - 5), (6, 44, None), (6, 44, 2 ** 44 - 17),
                                      (5, 30, 2 ** 30 - 35)])
def test_top_digit_fits_a_nibble_and_digits_reconstruct_the_residue(L, bits, q):
    if q is None:   # the BFVDefault primes the BASELINE configs use
        q = int(oracle.BFV_DEFAULT[4096][0]) if L == 5 else int(oracle.BFV_DEFAULT[8192][3])
    assert q <= 2 ** (8 * (L - 1) + 4)
    rng = np.random.default_rng(L * 100 + bits)
    vm = vmax(L)
    edge = [0, 1, q - 1, q - 2, q // 2, q // 2 + 1, vm, vm + 1, vm - 1, vm + 2, 7 * 256 ** (L - 1), 7 * 256 ** (L - 1) - 1,
            2 ** (bits - 1), 2 ** (bits - 1) - 1]
    xs = np.array([e for e in edge if 0 <= e < q] + [int(v) for v in rng.integers(0, q, size=20000)], dtype=np.uint64)
    d = to_digits_top4(xs, q, L)
    for a in range(L - 1):
        assert all(-128 <= int(v) <= 127 for v in d[a])
    assert all(-8 <= int(v) <= 7 for v in d[L - 1]), (min(d[L - 1]), max(d[L - 1]))
    value = sum(d[a] * 256 ** a for a in range(L))
    assert all((int(v) - int(x)) % q == 0 for v, x in zip(value, xs))
    # |centred value| stays below 9/16 of the digit range: the int32 accumulators of the digit products keep their bound
    assert all(abs(int(v)) <= 9 * 256 ** (L - 1) for v in value)


def test_the_two_centred_ranges_span_exactly_the_nibble_range():
    for L in (5, 6, 7):
        p = 256 ** (L - 1)
        vmin = -8 * p - 128 * ((p - 1) // 255)
        assert vmax(L) - vmin + 1 == 16 * p            # 2^(8 (L - 1) + 4) values: every residue of such a modulus fits


def test_nibble_row_pack_and_unpack_round_trip():
    rng = np.random.default_rng(7)
    for _ in range(2000):
        row = [int(v) for v in rng.integers(-8, 8, size=16)]
        w0, w1 = pack_top4(row)
        assert 0 <= w0 < 2 ** 32 and 0 <= w1 < 2 ** 32
        assert expand_top4(w0, w1) == row
    for row in ([-8] * 16, [7] * 16, list(range(-8, 8))):
        assert expand_top4(*pack_top4(row)) == row
