"""The ProcessRequest known answers of the reference's server_test.cpp (:97-272) -- integer database,
plaintext-level results `int_db[i] * next_power_two(n)` -- against the CPU oracle (CPU tests) and against the
GPU server with the product client decoding (GPU tests).  These pin the expansion scale factor, the
multi-ciphertext split, the batch order, the zero query and the two-dimensional reply format."""
import numpy as np
import pytest

import oracle
import pir_amd
from oracle.client import Client
from pir_amd import parameters as P

N = 4096          # server_test.cpp:58
ELEM_SIZE = 7680  # server_test.cpp:59: one item per plaintext


def integer_encode(value, t):
    """seal::IntegerEncoder::encode(int64), base 2."""
    mag, coeffs = abs(value), []
    while mag:
        coeffs.append((1 if value > 0 else t - 1) if mag & 1 else 0)
        mag >>= 1
    return coeffs or [0]


def integer_decode(pt, t):
    """seal::IntegerEncoder::decode_int64."""
    acc = 0
    for c in reversed([int(x) for x in pt]):
        acc = 2 * acc + (c - t if c >= (t + 1) // 2 else c)
    return acc


def gen_int_db(n, seed=42):
    """PIRTestingBase::GenerateIntDB (test_base.cpp:67-78): 6 random bytes per entry."""
    rng = np.random.default_rng(seed)
    return [int(v) for v in rng.integers(0, 1 << 48, size=n, dtype=np.uint64)]


class Fixture:
    """PIRServerTestBase::SetUpDBImpl (server_test.cpp:62-78) on the oracle."""

    def __init__(self, dbsize, dimensions=1):
        self.params = oracle.create_pir_parameters(dbsize, ELEM_SIZE, dimensions, N=N, plain_bits=20)
        assert self.params.items_per_plaintext == 1 and self.params.num_pt == dbsize
        self.orc = oracle.Oracle.from_params(self.params)
        self.t = self.params.t
        self.int_db = gen_int_db(dbsize)
        self.rows = [integer_encode(v, self.t) for v in self.int_db]
        self.db_ntt = self.orc.db_from_coeffs(self.rows)
        self.client = Client(self.orc, seed=5)
        self.keys = self.client.galois_keys()

    def query(self, plaintexts):
        return np.stack([self.client.encrypt(pt) for pt in plaintexts])

    def one_hot(self, entries, n_cts=1):
        pts = [np.zeros(N, dtype=np.uint64) for _ in range(n_cts)]
        for ct, slot, val in entries:
            pts[ct][slot] = val
        return self.query(pts)

    def decode(self, reply):
        if len(self.params.dimensions) == 1:
            return integer_decode(self.client.decrypt(reply[0]), self.t)
        return integer_decode(self.client.process_reply(self.params, reply), self.t)


# (name, dbsize, dims, query entries [(ct, slot, value or "minv")], n query cts, expected(fixture))
CASES = [
    ("SingleCT :97-120", 10, 1, [(0, 7, 1)], 1, lambda f: f.int_db[7] * P.next_power_two(10)),
    ("MultiCT :122-150", 5000, 1, [(1, 4200 - N, 1)], 2, lambda f: f.int_db[4200] * P.next_power_two(5000 - N)),
    ("ZeroInput :183-207", 10, 1, [], 1, lambda f: 0),
    ("2Dim :209-262", 82, 2, [(0, 4, "minv"), (0, 16, "minv")], 1, lambda f: f.int_db[42]),
]


def build_query(f, entries, n_cts):
    minv = pow(P.next_power_two(sum(f.params.dimensions)), -1, f.t)
    return f.one_hot([(ct, slot, minv if val == "minv" else val) for ct, slot, val in entries], n_cts)


@pytest.mark.parametrize("name,dbsize,dims,entries,n_cts,expected", CASES, ids=[c[0] for c in CASES])
def test_oracle_process_request_known_answers(name, dbsize, dims, entries, n_cts, expected):
    f = Fixture(dbsize, dims)
    if dims == 2:
        assert f.params.dimensions == [10, 9]
    q = build_query(f, entries, n_cts)
    rc, reply = f.orc.process_query(f.db_ntt, f.params.dimensions, q, f.keys)
    assert rc == 0 and reply.shape[0] == f.orc.reply_ct_count(dims)
    assert f.decode(reply) == expected(f)


def test_oracle_process_batch_request():
    # server_test.cpp:152-181: replies come back in query order
    f = Fixture(10)
    for idx in (3, 4, 5):
        rc, reply = f.orc.process_query(f.db_ntt, f.params.dimensions, f.one_hot([(0, idx, 1)]), f.keys)
        assert rc == 0 and f.decode(reply) == f.int_db[idx] * P.next_power_two(10)


# ---------------------------------------------------------------- the same on the GPU server

def gpu_server(f):
    from gpu_helpers import to_product_params
    pp = to_product_params(f.params)
    db = pir_amd.PIRDatabase.Create(pp)
    db.populate_coeffs(f.rows)
    srv = pir_amd.PIRServer.Create(db, pp)
    srv.set_galois_keys(f.keys)
    return db, srv


@pytest.mark.gpu
@pytest.mark.parametrize("name,dbsize,dims,entries,n_cts,expected", CASES, ids=[c[0] for c in CASES])
def test_gpu_process_request_known_answers(name, dbsize, dims, entries, n_cts, expected):
    f = Fixture(dbsize, dims)
    db, srv = gpu_server(f)
    q = build_query(f, entries, n_cts)
    reply = srv.process_query(q)
    rc, exp = f.orc.process_query(f.db_ntt, f.params.dimensions, q, f.keys)
    assert np.array_equal(reply, exp)
    assert f.decode(reply) == expected(f)
    db.close()


@pytest.mark.gpu
def test_gpu_process_batch_request_through_the_wire():
    """TestProcessBatchRequest (server_test.cpp:152-181) on serialized protos; the product client decodes."""
    import seal_wire as W
    f = Fixture(10)
    db, srv = gpu_server(f)
    indexes = [3, 4, 5]
    queries = [f.one_hot([(0, i, 1)]) for i in indexes]
    o = f.orc
    gk = W.save_galois_keys(f.keys, N, W.parms_id(N, o.moduli, o.t))
    resp = srv.ProcessRequest(W.save_request(queries, gk, W.parms_id(N, o.moduli[: o.k], o.t)))
    replies = W.load_response(resp)
    assert len(replies) == 3
    for i, rep in zip(indexes, replies):
        assert f.decode(rep) == f.int_db[i] * P.next_power_two(10)
    db.close()
