"""Pins the CPU oracle on the plaintext-level known answers of the reference's own
tests.  Each table is transcribed test DATA (inputs and
expected outputs) from the cited reference test; the ciphertexts are produced by
the oracle's CPU client with fresh keys, exactly as the reference tests do."""
import numpy as np
import pytest

import oracle
from oracle.client import Client
from conftest import parse_poly

N = 4096
T20 = 0xFC001


@pytest.fixture(scope="module")
def env():
    # server_test.cpp:60-84: POLY_MODULUS_DEGREE 4096, 20-bit plain modulus, BFVDefault
    t = oracle.plain_modulus_batching(N, 20)
    assert t == T20
    o = oracle.Oracle(N, oracle.BFV_DEFAULT[N], t)
    c = Client(o, seed=1234)
    return o, c


# server_test.cpp:291-305
SUBSTITUTIONS = [
    ("42", 3, "42"), ("1x^1", 5, "1x^5"), ("6x^2", 3, "6x^6"),
    ("1x^1", N + 1, "FC000x^1"), ("1x^4", N + 1, "1x^4"),
    ("1x^8", N // 2 + 1, "1x^8"), ("1x^8", N // 4 + 1, "1x^8"),
    ("1x^8", N // 8 + 1, "FC000x^8"), ("77x^4095", 3, "77x^4093"),
    ("1x^4095", N + 1, "FC000x^4095"),
    ("4x^4 + 33x^3 + 222x^2 + 19x^1 + 42", N + 1, "4x^4 + FBFCEx^3 + 222x^2 + FBFE8x^1 + 42"),
]


@pytest.mark.parametrize("inp,power,expected", SUBSTITUTIONS)
def test_substitute_examples(env, inp, power, expected):
    o, c = env
    ct = c.encrypt(parse_poly(inp, N))
    rc, out = o.apply_galois_ct(ct, power, c.galois_key(power))
    assert rc == 0
    assert (c.decrypt(out) == parse_poly(expected, N)).all()


def test_substitute_missing_key_is_internal_error(env):
    # server.cpp:72-74: SEAL throws when the key is absent -> InternalError
    o, c = env
    rc, _ = o.apply_galois_ct(c.encrypt(parse_poly("1", N)), 3, None)
    assert rc == oracle.INTERNAL


# server_test.cpp:333-339
INV_POWERS = [("42x^1", 1, "42"), ("42x^42", 41, "42x^1"),
              ("1x^4 + 1x^3 + 1x^1", 1, "1x^3 + 1x^2 + 1"),
              ("1x^16 + 1x^12 + 1x^8", 4, "1x^12 + 1x^8 + 1x^4")]


@pytest.mark.parametrize("inp,k,expected", INV_POWERS)
def test_multiply_inverse_power_x(env, inp, k, expected):
    o, c = env
    out = o.multiply_inverse_power_of_x(c.encrypt(parse_poly(inp, N)), k)
    assert (c.decrypt(out) == parse_poly(expected, N)).all()


# server_test.cpp:376-383
EXPANSIONS = [("1", ["2", "0"]), ("1x^1", ["0", "2"]),
              ("3x^3 + 2x^2 + 1x^1 + 42", ["108", "4", "8", "C"]),
              ("1x^5", ["0", "0", "0", "0", "0", "8"])]


@pytest.fixture(scope="module")
def gal_keys(env):
    o, c = env
    return c.galois_keys()


@pytest.mark.parametrize("inp,expected", EXPANSIONS)
def test_oblivious_expansion_examples(env, gal_keys, inp, expected):
    o, c = env
    rc, res = o.oblivious_expansion(c.encrypt(parse_poly(inp, N)), len(expected), gal_keys)
    assert rc == 0 and res.shape[0] == len(expected)
    for ct, e in zip(res, expected):
        assert (c.decrypt(ct) == parse_poly(e, N)).all()


def test_expansion_too_many_items_is_invalid_argument(env, gal_keys):
    o, c = env   # server.cpp:111-114
    rc, _ = o.oblivious_expansion(c.encrypt(parse_poly("1", N)), N + 1, gal_keys)
    assert rc == oracle.INVALID_ARGUMENT


# server_test.cpp:423-428 (num_items, index, expected scale); the 4096/5000-item
# cases run 4095+ key switches on the CPU oracle, so the largest are kept to one.
MULTI_CT = [(100, 42, 128), (100, 0, 128), (100, 99, 128), (5000, 4200, 1024)]


@pytest.mark.parametrize("num_items,index,value", MULTI_CT[:3])
def test_expansion_multi_ct(env, gal_keys, num_items, index, value):
    o, c = env
    n_ct = num_items // N + 1
    cts = []
    for i in range(n_ct):
        pt = np.zeros(N, dtype=np.uint64)
        if index // N == i:
            pt[index % N] = 1
        cts.append(c.encrypt(pt))
    rc, res = o.oblivious_expansion_multi(np.stack(cts), num_items, gal_keys)
    assert rc == 0 and res.shape[0] == num_items
    for i in range(num_items):
        d = c.decrypt(res[i])
        assert int(d[0]) == (value if i == index else 0)
        assert not d[1:].any()


def test_expansion_multi_ct_count_mismatch(env, gal_keys):
    o, c = env  # server.cpp:154-158: needs total/N + 1 ciphertexts even when total % N == 0
    ct = c.encrypt(parse_poly("1", N))
    rc, _ = o.oblivious_expansion_multi(np.stack([ct]), N, gal_keys)
    assert rc == oracle.INVALID_ARGUMENT


def test_expansion_ratio_and_reencode_roundtrip():
    # ct_reencoder_test.cpp:77-79: ExpansionRatio()==4 at N=4096, t=20 bits
    o = oracle.Oracle(N, oracle.BFV_DEFAULT[N], T20)
    assert o.expansion_ratio() == 4
    rng = np.random.default_rng(5)
    ct = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in o.moduli[:2]]) for _ in range(2)])
    pts = o.reencode(ct)
    assert pts.shape == (8, N) and int(pts.max()) < (1 << 19)
    assert (o.redecode(pts) == ct).all()
    # benchmark parameters (24-bit t): ceil(36/23) = 2 per prime
    o24 = oracle.Oracle(N, oracle.BFV_DEFAULT[N], oracle.plain_modulus_batching(N, 24))
    assert o24.expansion_ratio() == 4
