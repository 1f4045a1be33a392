"""The algebra behind the NTT-domain last expansion level (ks_last_ntt_kernel), checked with
the CPU oracle's own transforms, and the index maps the device code relies on.

  * sigma_g on NTT-form data is the permutation pi_g(P) = br(((2 br(P) + 1) g mod 2N - 1) / 2) of the SEAL
    (bit-reversed) positions:  NTT(sigma_g(a))[P] = NTT(a)[pi_g(P)]            (galois_ntt_slot in ntt_kernels.hip)
  * multiplying by x^(-s) is a dyadic product with X = NTT(-x^(N-s))              (xpow_table in ctx.hip)
  * hence, with lo = a + g and hi = x^(-s) (a - g) (reference server.cpp:137-141):
        NTT(lo) = NTT(a) + NTT(g),   NTT(hi) = X (.) (NTT(a) - NTT(g))
  * TwSoA<LOGN> (ntt_core.h): the slot-major LDS copy of a twiddle table holds every table entry 1 .. N-1 exactly once
"""
import numpy as np
import pytest

import oracle


def _br(v, bits):
    return int(format(v, "0%db" % bits)[::-1], 2)


def pi_g(P, g, logN):
    N = 1 << logN
    e = ((2 * _br(P, logN) + 1) * g) % (2 * N)
    return _br((e - 1) // 2, logN)


@pytest.fixture(scope="module")
def orc():
    N = 2048
    moduli = oracle.coeff_modulus_create(N, [36, 36, 37])
    return oracle.Oracle(N, moduli, oracle.plain_modulus_batching(N, 20))


@pytest.mark.parametrize("level", [0, 1, 4, 7, 10])
def test_galois_is_a_permutation_of_ntt_positions(orc, level):
    N, logN = orc.N, orc.logN
    g = (N >> level) + 1                       # the expansion's Galois elements (server.cpp:117-118)
    rng = np.random.default_rng(level)
    for mi in range(orc.k):
        a = rng.integers(0, orc.moduli[mi], N, dtype=np.uint64)
        lhs = orc.ntt_fwd(mi, orc.apply_galois_poly(mi, a, g))
        A = orc.ntt_fwd(mi, a)
        perm = np.array([pi_g(P, g, logN) for P in range(N)])
        assert np.array_equal(lhs, A[perm])
        # and the inverse permutation is pi of g^-1 mod 2N (how A_1 is recovered from the digit kernel's output)
        ginv = pow(g, -1, 2 * N)
        inv = np.array([pi_g(P, ginv, logN) for P in range(N)])
        assert np.array_equal(perm[inv], np.arange(N))


@pytest.mark.parametrize("level", [0, 3, 9])
def test_monomial_and_tree_butterfly_commute_with_the_transform(orc, level):
    N = orc.N
    s = 1 << level
    rng = np.random.default_rng(100 + level)
    for mi in range(orc.k):
        q = orc.moduli[mi]
        a = rng.integers(0, q, N, dtype=np.uint64)
        gpoly = rng.integers(0, q, N, dtype=np.uint64)
        mono = np.zeros(N, dtype=np.uint64)
        mono[N - s] = q - 1                                  # x^(-s) = -x^(N-s)
        X = orc.ntt_fwd(mi, mono)
        # reference order of operations: coefficient-domain butterfly, then the lazy forward transform
        lo = orc.poly_add(mi, a, gpoly)
        hi = orc.negacyclic_shift_poly(mi, orc.poly_sub(mi, a, gpoly), 2 * N - s)
        A, G = orc.ntt_fwd(mi, a), orc.ntt_fwd(mi, gpoly)
        assert np.array_equal(orc.ntt_fwd(mi, lo), orc.poly_add(mi, A, G))
        assert np.array_equal(orc.ntt_fwd(mi, hi), orc.dyadic_mul(mi, X, orc.poly_sub(mi, A, G)))


def _twsoa(logN):
    """Python restatement of TwSoA<LOGN> (ntt_core.h): returns {table index: LDS word}."""
    NT = 1 << (logN - 4)

    def next_lb(lb):
        return lb - 4 if lb >= 4 else 0

    passes = []                                              # (LB, RHI) in fwd_chain's order
    lb, rhi = logN - 4, 3
    while True:
        passes.append((lb, rhi))
        if lb == 0:
            break
        rhi = 3 if lb >= 4 else lb - 1
        lb = next_lb(lb)
    where, off = {}, 0
    for lb, rhi in passes:
        w0 = (8 >> rhi) - 1
        width = NT >> lb
        for rb in range(rhi, -1, -1):
            mm = 1 << (logN - 1 - (lb + rb))
            for g in range(8 >> rb):
                for outer in range(width):
                    where[mm + (outer << (3 - rb)) + g] = off + ((8 >> rb) - 1 + g - w0) * width + outer
        off += (16 - (8 >> rhi)) * width
    return where, off


@pytest.mark.parametrize("logN", [11, 12, 13])
def test_slot_major_twiddle_layout_is_a_bijection(logN):
    where, words = _twsoa(logN)
    N = 1 << logN
    assert words == N - 1
    assert sorted(where) == list(range(1, N))                # every twiddle the transform uses
    assert sorted(where.values()) == list(range(N - 1))      # each LDS word exactly once
