"""Full round trip on the GPU with product code only on both sides (correctness_test.cpp:95-113):
pir_amd.PIRClient.CreateRequest -> pir_amd.PIRServer.ProcessRequest (libpirgpu, wire level) ->
PIRClient.ProcessResponse.  The oracle takes no part; the expected values are the database items."""
import numpy as np
import pytest

import pir_amd
from pir_amd import parameters as P

from pir_fixtures import generate_test_db

pytestmark = pytest.mark.gpu

# correctness_test.cpp:118-147 (N, plain bits, elem size, bits_per_coeff, dbsize, d, indexes)
CASES = [
    (4096, 24, 64, 0, 10, 1, [0]),
    (4096, 24, 64, 0, 1000, 1, [42, 999]),
    (4096, 24, 288, 0, 3000, 2, [0, 1234, 2999]),
    (4096, 24, 288, 0, 3200, 2, [(173 * i + 5) % 3200 for i in range(19)]),   # one request, 19 queries: batch pipeline
    (4096, 24, 64, 10, 1500, 2, [7, 1499]),
    (4096, 20, 64, 0, 500, 3, [321]),
    (8192, 24, 256, 0, 2000, 2, [5, 1999]),
]


@pytest.mark.parametrize("N,bits,elem,bpc,dbsize,d,indexes", CASES)
def test_client_server_round_trip(N, bits, elem, bpc, dbsize, d, indexes):
    enc = P.generate_encryption_params(N, bits)
    pp = P.create_pir_parameters(dbsize, elem, d, enc, False, bpc)
    raw = generate_test_db(dbsize, elem)
    db = pir_amd.PIRDatabase.Create(pp, raw)
    server = pir_amd.PIRServer.Create(db, pp)
    client = pir_amd.PIRClient.Create(pp, seed=b"gpu-rt")
    request = client.CreateRequest(indexes)
    response = server.ProcessRequest(request)
    assert client.ProcessResponse(indexes, response) == [raw[i].tobytes() for i in indexes]
    # residue-level halves of the same calls
    server.set_galois_keys(client.galois_keys())
    reply = server.process_query(client.create_query_for(indexes[-1]))
    assert reply.shape[0] == client.reply_ct_count
    pt = client.process_reply(reply)
    off = (indexes[-1] % pp.items_per_plaintext) * elem
    assert client.string_decode(pt, elem, off) == raw[indexes[-1]].tobytes()
    assert client.noise_budget(reply[0]) > 0


def test_two_clients_share_one_server():
    """Each request carries its own keys (server.cpp:46-48): interleaved clients get their own items."""
    enc = P.generate_encryption_params(4096, 24)
    pp = P.create_pir_parameters(2000, 128, 2, enc)
    raw = generate_test_db(2000, 128)
    server = pir_amd.PIRServer.Create(pir_amd.PIRDatabase.Create(pp, raw), pp)
    a = pir_amd.PIRClient.Create(pp, seed=b"a")
    b = pir_amd.PIRClient.Create(pp)           # keyed from the OS
    for client, idx in [(a, [3]), (b, [1999]), (a, [77, 78]), (b, [0])]:
        assert client.ProcessResponse(idx, server.ProcessRequest(client.CreateRequest(idx))) == \
            [raw[i].tobytes() for i in idx]


@pytest.mark.parametrize("log_items", [8, 10, 12, 14, 16])
def test_reference_benchmark_sizes_recover_the_item_and_match_the_oracle(log_items):
    """The reference's own benchmark cases (benchmark.cpp:17-23, 102-104: 2^8 .. 2^16 items of 288 bytes, d = 2, N = 4096,
    24-bit t, one query per request): the product client's request through the wire-level server, the item back -- the
    parameters leave a positive noise budget at every one of these sizes -- and the reply equal to the oracle's, bit for
    bit, on the same query ciphertext and keys."""
    import oracle
    from gpu_helpers import to_product_params
    n = 1 << log_items
    enc = P.generate_encryption_params(4096, 24)
    pp = P.create_pir_parameters(n, 288, 2, enc)
    raw = generate_test_db(n, 288, seed=100 + log_items)
    db = pir_amd.PIRDatabase.Create(pp, raw)
    server = pir_amd.PIRServer.Create(db, pp)
    client = pir_amd.PIRClient.Create(pp, seed=b"reference-sizes-%d" % log_items)
    idx = (n * 5) // 7
    response = server.ProcessRequest(client.CreateRequest([idx]))
    assert client.ProcessResponse([idx], response) == [raw[idx].tobytes()]
    # the oracle on THIS client's keys and one of its query ciphertexts
    op = oracle.create_pir_parameters(n, 288, 2, N=4096, plain_bits=24)
    assert list(op.dimensions) == list(pp.dimensions) and to_product_params(op).num_pt == pp.num_pt
    orc = oracle.Oracle.from_params(op)
    rc, db_ntt = orc.db_encode(raw.tobytes(), n, 288, op.items_per_plaintext, op.eff_bits_per_coeff, op.num_pt)
    assert rc == 0
    q = client.create_query_for(idx)
    keys = client.galois_keys()
    rc, want = orc.process_query(db_ntt, op.dimensions, q, keys)
    assert rc == 0
    server.set_galois_keys(keys)
    got = server.process_query(q)
    assert np.array_equal(got, want)
    assert client.noise_budget(got[0]) > 0
    db.close()
