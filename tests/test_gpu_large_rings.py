"""The reference's LARGE-RING cases on the GPU, bit for bit against the oracle (missing #3):

  * database_test.cpp:387-388 -- MultiplyMultiDimTest (8192, 20, 27, 3, 2) and (8192, 20, 117, 3, 17): N = 8192 with
    the FULL BFVDefault(8192) chain (k = 4 data primes of 43 / 43 / 44 / 44 bits + the 44-bit special prime), d = 3;
  * correctness_test.cpp:99 -- (8192, 42-bit t, 87 items, d = 2) through the whole query path;
  * NTT / substitute_power_x / oblivious_expansion at N = 8192 with k = 4 (test_gpu_parity.py uses k = 3 there).

Every reply is np.array_equal to oracle.process_query / db_multiply on the same inputs; the plaintext-level answers of
the reference's tests (the item comes back) are checked on top."""
import numpy as np
import pytest

import oracle
import pir_amd
from gpu_helpers import random_ct, random_key, to_product_params
from pir_fixtures import PirSetup

pytestmark = pytest.mark.gpu

N8 = 8192


def _server(s, shard=None):
    pp = to_product_params(s.params)
    db = pir_amd.PIRDatabase.Create(pp, shard=shard)
    db.populate(s.raw)
    srv = pir_amd.PIRServer.Create(db, pp) if shard is None else pir_amd.PIRServer(db, pp)
    srv.set_galois_keys(s.galois_keys)
    return db, srv


@pytest.fixture(scope="module")
def ring8k4():
    """N = 8192, the full default chain: k = 4."""
    s = PirSetup(10, 0, 1, N=N8, plain_bits=20)
    assert s.orc.k == 4 and [int(q).bit_length() for q in s.orc.moduli] == [43, 43, 44, 44, 44]
    db, srv = _server(s)
    yield s, db, srv
    db.close()


@pytest.mark.parametrize("dbsize,index", [(27, 2), (117, 17)])
def test_database_test_multiply_multi_dim_n8192_d3(dbsize, index):
    # database_test.cpp:349-388 MultiplyMultiDimTest CTDecomp: (8192, 20, dbsize, 3, desired_index), element size 0
    s = PirSetup(dbsize, 0, 3, N=N8, plain_bits=20)
    p = s.params
    assert s.orc.k == 4 and len(p.dimensions) == 3
    db, srv = _server(s)
    assert db.expansion_ratio() == s.orc.expansion_ratio()
    q = s.client.create_query_for(p, index)
    rc, sv = s.orc.oblivious_expansion_multi(q, p.dim_sum, s.galois_keys)
    assert rc == 0
    rc, exp = s.orc.db_multiply(s.db_ntt, p.dimensions, sv.copy())
    assert rc == 0
    got = db.multiply(sv)                                     # PIRDatabase::multiply on the same selection vector
    assert got.shape == exp.shape and np.array_equal(got, exp)
    assert np.array_equal(srv.process_query(q), exp)          # ... and the whole processQuery
    assert s.client.process_response(p, index, got) == s.item(index)
    # several queries through the batch pipeline: the group-wise d = 3 recursion at k = 4
    idx = [(index + 5 * i) % dbsize for i in range(5)]
    qs = np.stack([s.client.create_query_for(p, i) for i in idx])
    srv.set_concurrency(8)
    srv.stage_batch(qs)
    srv.run_batch()
    out = srv.fetch_batch()
    for i, qq in zip(idx, qs):
        rc, want = s.orc.process_query(s.db_ntt, p.dimensions, qq, s.galois_keys)
        assert rc == 0 and np.array_equal(out[idx.index(i)], want)
    db.close()


def test_correctness_test_n8192_t42_d2():
    # correctness_test.cpp:99: (8192, 42-bit plain modulus, 87 items, d = 2, indices 5 / 33 / 86)
    s = PirSetup(87, 0, 2, N=N8, plain_bits=42)
    p = s.params
    assert s.orc.k == 4
    db, srv = _server(s)
    for index in (5, 33, 86):
        q = s.client.create_query_for(p, index)
        rc, exp = s.orc.process_query(s.db_ntt, p.dimensions, q, s.galois_keys)
        assert rc == 0
        got = srv.process_query(q)
        assert np.array_equal(got, exp)
        assert s.client.process_response(p, index, got) == s.item(index)
    db.close()


def test_ntt_parity_n8192_k4(ring8k4):
    s, db, srv = ring8k4
    rng = np.random.default_rng(8192)
    cts = random_ct(s.orc, rng, 3)
    fwd = srv.ntt_forward(cts)
    assert np.array_equal(fwd, np.stack([s.orc.ct_ntt_fwd(c) for c in cts]))
    assert np.array_equal(srv.ntt_inverse(fwd), cts)
    kl = np.empty((2, s.orc.k + 1, N8), dtype=np.uint64)       # key level: q_0..q_3, p
    for i in range(s.orc.k + 1):
        kl[:, i, :] = rng.integers(0, s.orc.moduli[i], size=(2, N8), dtype=np.uint64)
    fk = srv.ntt_forward(kl, key_level=True)
    for b in range(2):
        for i in range(s.orc.k + 1):
            assert np.array_equal(fk[b, i], s.orc.ntt_fwd(i, kl[b, i]))
    assert np.array_equal(srv.ntt_inverse(fk, key_level=True), kl)


@pytest.mark.parametrize("power", [3, N8 + 1, N8 // 4 + 1, 2 * N8 - 1])
def test_substitute_parity_n8192_k4(ring8k4, power):
    s, db, srv = ring8k4
    rng = np.random.default_rng(power)
    ct = random_ct(s.orc, rng)[0]
    key = random_key(s.orc, rng)                               # uniform residues: every rounding boundary of the key switch
    srv.set_galois_keys({power: key})
    rc, exp = s.orc.apply_galois_ct(ct, power, key)
    assert rc == 0
    assert np.array_equal(srv.substitute_power_x_inplace(ct.copy(), power), exp)
    srv.set_galois_keys(s.galois_keys)


@pytest.mark.parametrize("n", [1, 3, 10, 16])   # (the fixture's workspace is sized for its 10-plaintext database: m <= 16)
def test_expansion_parity_n8192_k4(ring8k4, n):
    s, db, srv = ring8k4
    pt = np.zeros(N8, dtype=np.uint64)
    pt[:4] = [42, 1, 2, 3]
    pt[n - 1] += 7
    ct = s.client.encrypt(pt)
    rc, exp = s.orc.oblivious_expansion(ct, n, s.galois_keys)
    assert rc == 0
    assert np.array_equal(srv.oblivious_expansion(ct, n), exp)
