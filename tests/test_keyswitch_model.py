"""Pins the C oracle (oracle/pir_oracle.c) against an INDEPENDENT big-integer model (tests/keyswitch_model.py)
written from SURVEY.md App. A.2-A.5 instead of from the oracle's source: minimal-root choice and NTT
ordering, Galois index map, key switching with the special-prime divide-and-round (both the per-modulus
formula with +floor(p/2) and a CRT big-integer floor((X + p/2)/p)), the expansion tree of server.cpp:105-146
and the recursive multiply of database.cpp:170-258 -- bit for bit at toy ring degrees (N = 16, 32).

This does not replace a SEAL-produced vector (none exists in this image; tools/check_external_pair.py takes
one when a SEAL machine provides it), but it removes "the GPU equals the builder's one reading of SEAL":
two separately written formulations of the published algorithm now have to agree on every residue."""
import numpy as np
import pytest

import oracle
from keyswitch_model import Model, Ring, minimal_primitive_root


def primes_1_mod(m, bits, count, skip=0):
    out, v = [], (1 << bits) - m + 1
    while len(out) < count + skip:
        if oracle.is_prime(v):
            out.append(v)
        v -= m
    return out[skip:]


CASES = [
    # N, data-prime bits, special bits, k, t
    (16, 30, 31, 2, 257),
    (16, 45, 46, 3, 65537),
    (32, 36, 37, 2, 12289),
    (32, 20, 28, 1, 193),
    (16, 59, 60, 2, 65537),      # largest moduli the device code accepts (< 2^61) -> integer NTT flavour territory
]


def setup(N, qb, pb, k, t, seed):
    moduli = primes_1_mod(2 * N, qb, k) + primes_1_mod(2 * N, pb, 1, skip=1 if pb == qb else 0)
    orc = oracle.Oracle(N, moduli, t)
    mdl = Model(N, moduli)
    rng = np.random.default_rng(seed)
    return moduli, orc, mdl, rng


def rand_ct(rng, moduli, k, N, n=1):
    out = np.empty((n, 2, k, N), dtype=np.uint64)
    for j in range(k):
        out[:, :, j, :] = rng.integers(0, moduli[j], size=(n, 2, N), dtype=np.uint64)
    return out


def rand_key(rng, moduli, k, N):
    key = np.empty((k, 2, k + 1, N), dtype=np.uint64)
    for i in range(k + 1):
        key[:, :, i, :] = rng.integers(0, moduli[i], size=(k, 2, N), dtype=np.uint64)
    return key


def as_np(x):
    return np.array(x, dtype=object).astype(np.uint64)


@pytest.mark.parametrize("N,qb,pb,k,t", CASES)
def test_minimal_root_and_ntt_order(N, qb, pb, k, t):
    moduli, orc, mdl, rng = setup(N, qb, pb, k, t, 1)
    for i, q in enumerate(moduli):
        assert orc.psi(i) == minimal_primitive_root(2 * N, q) == mdl.rings[i].psi
        a = rng.integers(0, q, size=N, dtype=np.uint64)
        want = mdl.rings[i].ntt(a.tolist())
        assert orc.ntt_fwd(i, a).tolist() == want
        assert orc.ntt_inv(i, np.array(want, dtype=np.uint64)).tolist() == a.tolist()
        assert mdl.rings[i].intt(want) == a.tolist()


@pytest.mark.parametrize("N,qb,pb,k,t", CASES)
def test_divide_round_special_rns_equals_bigint_equals_oracle(N, qb, pb, k, t):
    moduli, orc, mdl, rng = setup(N, qb, pb, k, t, 2)
    p = moduli[-1]
    S = [rng.integers(0, q, size=N, dtype=np.uint64).tolist() for q in moduli]
    # boundary residues of the special prime: 0, p-1, floor(p/2), floor(p/2) +- 1 (where a wrong rounding flips)
    for c, v in enumerate([0, p - 1, p // 2, p // 2 + 1, p // 2 - 1]):
        S[k][c] = v
    a = mdl.mod_down_rns(S)
    b = mdl.mod_down_bigint(S)
    assert a == b
    got = orc.divide_round_special(np.array(S, dtype=np.uint64))
    assert got.tolist() == a


@pytest.mark.parametrize("N,qb,pb,k,t", CASES)
def test_apply_galois_key_switch(N, qb, pb, k, t):
    moduli, orc, mdl, rng = setup(N, qb, pb, k, t, 3)
    for g in [3, N + 1, N // 2 + 1, 2 * N - 1]:
        ct = rand_ct(rng, moduli, k, N)[0]
        key = rand_key(rng, moduli, k, N)
        rc, got = orc.apply_galois_ct(ct, g, key)
        assert rc == 0
        want = mdl.apply_galois_ct(ct.tolist(), g, key.tolist())
        want_big = mdl.apply_galois_ct(ct.tolist(), g, key.tolist(), bigint=True)
        assert want == want_big
        assert got.tolist() == want, g


@pytest.mark.parametrize("N,qb,pb,k,t,n", [(16, 30, 31, 2, 257, 16), (16, 30, 31, 2, 257, 5), (32, 36, 37, 2, 12289, 11),
                                            (16, 45, 46, 3, 65537, 7)])
def test_oblivious_expansion(N, qb, pb, k, t, n):
    moduli, orc, mdl, rng = setup(N, qb, pb, k, t, 4)
    ct = rand_ct(rng, moduli, k, N)[0]
    keys = {(N >> j) + 1: rand_key(rng, moduli, k, N) for j in range(N.bit_length() - 1)}
    rc, got = orc.oblivious_expansion(ct, n, keys)
    assert rc == 0
    want = mdl.oblivious_expansion(ct.tolist(), n, {g: key.tolist() for g, key in keys.items()})
    assert got.tolist() == want


@pytest.mark.parametrize("N,qb,pb,k,t,dims,P", [(16, 30, 31, 2, 257, [5], 5), (16, 30, 31, 2, 257, [3, 3], 8),
                                                (32, 36, 37, 2, 12289, [2, 3], 6), (16, 45, 46, 3, 65537, [2, 2, 2], 7)])
def test_db_multiply(N, qb, pb, k, t, dims, P):
    moduli, orc, mdl, rng = setup(N, qb, pb, k, t, 5)
    rows = [rng.integers(0, t, size=int(rng.integers(1, N + 1)), dtype=np.uint64) for _ in range(P)]
    db_ntt = orc.db_from_coeffs(rows)
    sv = rand_ct(rng, moduli, k, N, sum(dims))
    assert orc.expansion_ratio() == mdl.expansion_ratio(t)
    rc, got = orc.db_multiply(db_ntt, dims, sv.copy())
    assert rc == 0
    want = mdl.db_multiply([r.tolist() for r in rows], dims, sv.tolist(), t)
    assert got.tolist() == want
