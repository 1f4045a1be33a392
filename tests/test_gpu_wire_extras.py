"""Wire-level behaviours added in round 2, on the GPU through pirgpu_process_request:
seed-compressed key objects (what a reference client sends) against their expanded twins, RelinKeys parsed and
validated like server.cpp:53-58, SEAL's "result ciphertext is transparent" status (database.cpp:313-315) and its
opt-out, and two threads serving different clients on ONE context (the request is one critical section)."""
import threading

import numpy as np
import pytest

import oracle
import pir_amd
from pir_amd import parameters as P
from gpu_helpers import random_ct, random_key, to_product_params
from pir_fixtures import PirSetup, generate_test_db

pytestmark = pytest.mark.gpu


def _split(request):
    import seal_wire as W
    fields = W._parse(request)
    return ([p for n, p in fields if n == 1], [p for n, p in fields if n == 2], [p for n, p in fields if n == 3])


def test_seeded_and_expanded_keys_give_the_same_response():
    enc = P.generate_encryption_params(4096, 24)
    pp = P.create_pir_parameters(3000, 288, 2, enc)
    raw = generate_test_db(3000, 288)
    server = pir_amd.PIRServer.Create(pir_amd.PIRDatabase.Create(pp, raw), pp)
    client = pir_amd.PIRClient.Create(pp, seed=b"seeded")
    idx = [11, 2999]
    seeded = client.CreateRequest(idx)
    client.set_seeded_keys(False)
    expanded = client.CreateRequest(idx)
    assert len(seeded) < 0.6 * len(expanded)
    r1, r2 = server.ProcessRequest(seeded), server.ProcessRequest(expanded)
    # the query ciphertexts differ (fresh encryption randomness), the keys are the same keys: both decode
    assert client.ProcessResponse(idx, r1) == client.ProcessResponse(idx, r2) == [raw[i].tobytes() for i in idx]
    # same queries, the two key encodings: byte-identical responses
    import seal_wire as W
    q, g_seeded, relin = _split(seeded)
    _, g_expanded, _ = _split(expanded)

    def rebuild(galois):
        out = b"".join(W._field(1, bytes(x)) for x in q) + W._field(2, bytes(galois))
        return out + W._field(3, bytes(relin[0]))
    assert server.ProcessRequest(rebuild(g_seeded[0])) == server.ProcessRequest(rebuild(g_expanded[0]))


def test_relin_keys_are_validated():
    enc = P.generate_encryption_params(4096, 24)
    pp = P.create_pir_parameters(500, 64, 1, enc)
    raw = generate_test_db(500, 64)
    server = pir_amd.PIRServer.Create(pir_amd.PIRDatabase.Create(pp, raw), pp)
    client = pir_amd.PIRClient.Create(pp, seed=b"relin")
    import seal_wire as W
    request = client.CreateRequest([7])
    q, g, relin = _split(request)
    assert len(relin) == 1 and len(relin[0]) > 1000                 # the reference client always sends them
    base = b"".join(W._field(1, bytes(x)) for x in q) + W._field(2, bytes(g[0]))
    assert client.ProcessResponse([7], server.ProcessRequest(base)) == [raw[7].tobytes()]      # absent: fine
    for bad in (bytes(relin[0][:-9]), b"\x00" * 64, bytes(relin[0][:16]) + b"\xff" * 40):
        with pytest.raises(pir_amd.PirGpuError) as e:
            server.ProcessRequest(base + W._field(3, bad))
        assert e.value.code == pir_amd.StatusCode.INVALID_ARGUMENT   # server.cpp:53-58 via serialization.h:113-115


def test_transparent_ciphertext_status_and_opt_out():
    """An identically-zero database plaintext: SEAL throws from multiply_plain, the reference returns Internal
    (database.cpp:313-315); so do the oracle and, by default, the GPU path.  With the opt-out the defined reply."""
    s = PirSetup(6, 0, 1, N=4096, plain_bits=20)
    rng = np.random.default_rng(4)
    rows = [rng.integers(0, s.orc.t, size=4096, dtype=np.uint64) for _ in range(6)]
    rows[4][:] = 0
    pp = to_product_params(s.params)
    db = pir_amd.PIRDatabase.Create(pp)
    db.populate_coeffs(rows)
    srv = pir_amd.PIRServer.Create(db, pp)
    srv.set_galois_keys(s.galois_keys)
    q = s.client.create_query_for(s.params, 2)
    db_ntt = s.orc.db_from_coeffs(rows)
    rc, _ = s.orc.process_query(db_ntt, s.params.dimensions, q, s.galois_keys)
    assert rc == oracle.INTERNAL
    with pytest.raises(pir_amd.PirGpuError) as e:
        srv.process_query(q)
    assert e.value.code == pir_amd.StatusCode.INTERNAL and "transparent" in e.value.message
    with pytest.raises(pir_amd.PirGpuError) as e:
        srv.process_batch(np.stack([q, q]), n_workers=2)
    assert e.value.code == pir_amd.StatusCode.INTERNAL
    db.set_transparent_policy(True)
    reply = srv.process_query(q)
    assert np.array_equal(s.client.decrypt(reply[0]), rows[2])
    # reloading the plaintext with non-zero content clears the condition
    db.set_transparent_policy(False)
    rows[4][0] = 1
    db.populate_coeffs(rows)
    assert srv.process_query(q).shape[0] == 1
    db.close()


def test_concurrent_requests_on_one_context():
    """Two threads, two clients, one context: every response must decode under its own client's keys."""
    enc = P.generate_encryption_params(4096, 24)
    pp = P.create_pir_parameters(2000, 128, 2, enc)
    raw = generate_test_db(2000, 128)
    server = pir_amd.PIRServer.Create(pir_amd.PIRDatabase.Create(pp, raw), pp)
    clients = [pir_amd.PIRClient.Create(pp, seed=b"t%d" % i) for i in range(2)]
    errors = []

    def work(ci):
        try:
            c = clients[ci]
            for it in range(6):
                idx = [(37 * it + 11 * ci) % 2000, (5 * it + 900 * ci + 1) % 2000]
                items = c.ProcessResponse(idx, server.ProcessRequest(c.CreateRequest(idx)))
                if items != [raw[i].tobytes() for i in idx]:
                    errors.append((ci, it))
        except Exception as ex:            # noqa: BLE001
            errors.append((ci, repr(ex)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
