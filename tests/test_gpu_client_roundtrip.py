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
