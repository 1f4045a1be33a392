"""End-to-end checks of the CPU oracle: Client -> oracle server path -> Client, on the
decomposition-mode tuples of the reference's correctness_test.cpp:106-113 and the
multiply/dot-product tests of database_test.cpp."""
import numpy as np
import pytest

import oracle
from pir_fixtures import PirSetup

# correctness_test.cpp:106-113 (use_ct_mult=false rows): (N, plain bits, elem size, bits/coeff, dbsize, d, indexes)
CORRECTNESS = [
    (4096, 24, 0, 0, 10, 1, [0]),
    (4096, 24, 0, 10, 9, 2, [1, 5]),
    (4096, 24, 0, 6, 500, 2, [9, 125]),
    (4096, 24, 64, 10, 1200, 1, [0, 80, 81, 123, 777, 1199]),
    (4096, 24, 289, 10, 1200, 1, [0, 47, 777, 1199]),
]


@pytest.mark.parametrize("N,pbits,elem,bpc,dbsize,d,indexes", CORRECTNESS)
def test_correctness(N, pbits, elem, bpc, dbsize, d, indexes):
    s = PirSetup(dbsize, elem, d, N=N, plain_bits=pbits, bits_per_coeff=bpc)
    for idx in indexes[:3]:
        q = s.client.create_query_for(s.params, idx)
        rc, reply = s.orc.process_query(s.db_ntt, s.params.dimensions, q, s.galois_keys)
        assert rc == 0
        assert reply.shape[0] == s.orc.reply_ct_count(d)
        assert s.client.process_response(s.params, idx, reply) == s.item(idx)


def test_benchmark_shape_small():
    # benchmark.cpp:17-23 parameters (288 B items, d=2, N=4096, 24-bit t) at the smallest swept size 2^8
    s = PirSetup(256, 288, 2, N=4096, plain_bits=24)
    assert s.params.items_per_plaintext == 40 and s.params.dimensions == [3, 3]
    for idx in (0, 133, 255):
        q = s.client.create_query_for(s.params, idx)
        rc, reply = s.orc.process_query(s.db_ntt, s.params.dimensions, q, s.galois_keys)
        assert rc == 0 and reply.shape[0] == 8
        assert s.client.process_response(s.params, idx, reply) == s.item(idx)


def test_multiply_selection_vector_size_mismatch():
    # database_test.cpp:180-219 -> InvalidArgument
    s = PirSetup(10, 0, 1)
    sv = s.orc.new_ct(9)
    rc, _ = s.orc.db_multiply(s.db_ntt, s.params.dimensions, sv)
    assert rc == oracle.INVALID_ARGUMENT


def test_multiply_dot_product_plain_selection():
    # database_test.cpp:155-178: selection vector of encrypted constants -> dot product with the DB
    s = PirSetup(10, 0, 1, plain_bits=20)
    o, c = s.orc, s.client
    N, t = o.N, o.t
    rng = np.random.default_rng(3)
    dbc = [rng.integers(0, 1 << 10, N, dtype=np.uint64) for _ in range(10)]
    db = o.db_from_coeffs(dbc)
    sel = [int(x) for x in rng.integers(0, 8, 10)]
    sv = np.stack([c.encrypt(np.array([v], dtype=np.uint64)) for v in sel])
    rc, out = o.db_multiply(db, [10], sv)
    assert rc == 0 and out.shape[0] == 1
    exp = sum(v * d.astype(object) for v, d in zip(sel, dbc)) % t
    assert (c.decrypt(out[0]).astype(object) == exp).all()


def test_multi_dim_3():
    # database_test.cpp:387-388 style: d=3 recursion (N=4096 here to keep the CPU suite short)
    s = PirSetup(27, 0, 3, N=4096, plain_bits=24)
    assert s.params.dimensions == [3, 3, 3]
    idx = 14
    q = s.client.create_query_for(s.params, idx)
    rc, reply = s.orc.process_query(s.db_ntt, s.params.dimensions, q, s.galois_keys)
    assert rc == 0 and reply.shape[0] == 64
    assert s.client.process_response(s.params, idx, reply) == s.item(idx)
