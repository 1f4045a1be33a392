"""Serving several clients at once ("What's missing" #2).  In the reference the Galois keys are
locals of one ProcessRequest call (server.cpp:46-48): requests of different clients are independent.  Here the keys
of up to `capacity` clients stay resident (pirgpu_keyset_*), every query names its client's key set, and the queries
of ONE batch group may belong to different clients -- the key-switch kernels pick the key of each query's client.

Checked: 8 clients with different keys, one query each, expanded as one group, every reply bit-exact against the
oracle run with THAT client's keys (all arithmetic flavours; d = 1 and d = 2); pirgpu_process_requests == the
requests served one by one, byte for byte; alternating clients never re-upload; LRU eviction; a bad request inside
a batch; the speculative fingerprint match (same sampled fingerprint, different bytes must not reuse the keys)."""
import threading

import numpy as np
import pytest

import pir_amd
from pir_amd import parameters as P
from gpu_helpers import to_product_params
from oracle.client import Client
from pir_fixtures import PirSetup, generate_test_db

pytestmark = pytest.mark.gpu


def _server(s):
    pp = to_product_params(s.params)
    db = pir_amd.PIRDatabase.Create(pp, s.raw)
    return db, pir_amd.PIRServer.Create(db, pp)


@pytest.mark.parametrize("mode", ["default", "0", "2"])
@pytest.mark.parametrize("d,items,elem", [(2, 3000, 288), (1, 300, 288)], ids=["d2-mfma-scan", "d1"])
def test_eight_clients_with_different_keys_in_one_group(d, items, elem, mode, monkeypatch):
    if mode != "default":
        monkeypatch.setenv("PIRGPU_NTT_MODE", mode)
    s = PirSetup(items, elem, d, N=4096, plain_bits=24)
    p = s.params
    db, srv = _server(s)
    clients = [Client(s.orc, seed=1000 + i) for i in range(8)]
    keys = [c.galois_keys() for c in clients]
    slots = [srv.install_keyset(b"client-%d" % i, keys[i]) for i in range(8)]
    assert sorted(slots) == list(range(1, 9))
    idx = [(items - 1 - 311 * i) % items for i in range(8)]
    queries = np.stack([clients[i].create_query_for(p, idx[i]) for i in range(8)])
    srv.set_concurrency(8)                       # 8 workers = ONE group of 8 on one lane
    srv.stage_batch(queries)
    srv.set_batch_keysets(slots)
    srv.run_batch()
    got = srv.fetch_batch()
    for i in range(8):
        rc, exp = s.orc.process_query(s.db_ntt, p.dimensions, queries[i], keys[i])
        assert rc == 0
        assert np.array_equal(got[i], exp), "client %d" % i
        assert clients[i].process_response(p, idx[i], got[i]) == s.item(idx[i])
    # the wrong client's keys give a different reply (the test would not notice a kernel that ignored the slots)
    srv.stage_batch(queries)
    srv.set_batch_keysets(slots[1:] + slots[:1])
    srv.run_batch()
    wrong = srv.fetch_batch()
    assert not np.array_equal(wrong[0], got[0])
    rc, exp = s.orc.process_query(s.db_ntt, p.dimensions, queries[0], keys[1])
    assert np.array_equal(wrong[0], exp)
    # single-query entry points with a named slot
    srv.use_keyset(slots[5])
    assert np.array_equal(srv.process_query(queries[5]), got[5])
    srv.use_keyset(0)
    srv.set_galois_keys(keys[2])                # slot 0: the classic API
    assert np.array_equal(srv.process_query(queries[2]), got[2])
    assert srv.keyset_stats()["resident"] == 8
    # re-installing a resident client uploads nothing
    before = srv.keyset_stats()["key_uploads"]
    assert srv.install_keyset(b"client-3", keys[3]) == slots[3]
    assert srv.keyset_stats()["key_uploads"] == before
    db.close()


def _product_setup(items=2000, elem=128, d=2, n_clients=3):
    enc = P.generate_encryption_params(4096, 24)
    pp = P.create_pir_parameters(items, elem, d, enc)
    raw = generate_test_db(items, elem)
    server = pir_amd.PIRServer.Create(pir_amd.PIRDatabase.Create(pp, raw), pp)
    clients = [pir_amd.PIRClient.Create(pp, seed=b"mc%d" % i) for i in range(n_clients)]
    return pp, raw, server, clients


def test_process_requests_equals_one_by_one():
    pp, raw, server, clients = _product_setup(n_clients=5)
    wants = [[(97 * i + 3) % 2000] if i % 2 else [(97 * i + 3) % 2000, (411 * i + 9) % 2000] for i in range(5)]
    requests = [c.CreateRequest(w) for c, w in zip(clients, wants)]
    together = server.ProcessRequests(requests)
    assert [st for st, _ in together] == [0] * 5
    for c, w, (st, resp) in zip(clients, wants, together):
        assert c.ProcessResponse(w, resp) == [raw[i].tobytes() for i in w]
    st = server.keyset_stats()
    assert st["resident"] == 5 and st["evictions"] == 0
    uploads = st["key_uploads"]
    # the server is deterministic: served one by one (keys now resident) the same bytes come back
    for req, (_, resp) in zip(requests, together):
        assert server.ProcessRequest(req) == resp
    assert server.keyset_stats()["key_uploads"] == uploads          # nothing re-uploaded
    # a malformed request in the middle fails alone
    bad = requests[1][:-7]
    mixed = server.ProcessRequests([requests[0], bad, requests[2]])
    assert mixed[0] == together[0] and mixed[2] == together[2]
    assert mixed[1][0] == pir_amd.StatusCode.INVALID_ARGUMENT and mixed[1][1] is None
    assert server.request_errors[0] == "" and server.request_errors[2] == "" and server.request_errors[1] != ""
    # the two-halves form: ONE calling thread, two calls in flight (the second handed over before the first is waited
    # for); the same bytes come back, the malformed request still fails alone
    t1 = server.ProcessRequestsBegin(requests)
    t2 = server.ProcessRequestsBegin([requests[0], bad, requests[2]])
    assert server.ProcessRequestsEnd(t1) == together
    got2 = server.ProcessRequestsEnd(t2)
    assert got2[0] == together[0] and got2[2] == together[2] and got2[1] == (pir_amd.StatusCode.INVALID_ARGUMENT, None)
    assert server.request_errors[1] != "" and server.request_errors[0] == ""
    # a token is spent by End (a second End used to delete the library's call object twice) ...
    from pir_amd.server import PirGpuError
    with pytest.raises(PirGpuError) as e:
        server.ProcessRequestsEnd(t1)
    assert e.value.code == 9 and not t1.pending()
    # ... and a token that is dropped without End ends its call itself: the serving thread writes into the token's
    # arrays until then
    t3 = server.ProcessRequestsBegin(requests[:2])
    assert t3.pending()
    del t3
    import gc
    gc.collect()
    assert server.ProcessRequests(requests[:2]) == together[:2]


def test_alternating_clients_never_reupload_and_threads_are_combined():
    pp, raw, server, clients = _product_setup(n_clients=2)
    errors = []

    def work(ci):
        try:
            c = clients[ci]
            for it in range(8):
                idx = [(37 * it + 11 * ci) % 2000]
                items = c.ProcessResponse(idx, server.ProcessRequest(c.CreateRequest(idx)))
                if items != [raw[i].tobytes() for i in idx]:
                    errors.append((ci, it))
        except Exception as ex:            # noqa: BLE001
            errors.append((ci, repr(ex)))

    # strictly alternating on one thread first, then two threads at once
    for it in range(3):
        for ci in range(2):
            idx = [it * 5 + ci]
            assert clients[ci].ProcessResponse(idx, server.ProcessRequest(clients[ci].CreateRequest(idx))) == \
                [raw[idx[0]].tobytes()]
    n_keys = len(pir_amd.generate_galois_elts(4096))
    assert server.keyset_stats()["key_uploads"] == 2 * n_keys        # each client's keys went up exactly once
    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    st = server.keyset_stats()
    assert st["key_uploads"] == 2 * n_keys and st["evictions"] == 0 and st["resident"] == 2


def test_least_recently_used_key_set_is_evicted():
    pp, raw, server, clients = _product_setup(n_clients=3)
    server.set_keyset_capacity(2)
    n_keys = len(pir_amd.generate_galois_elts(4096))
    for rnd in range(2):
        for ci, c in enumerate(clients):
            idx = [100 * rnd + ci]
            assert c.ProcessResponse(idx, server.ProcessRequest(c.CreateRequest(idx))) == [raw[idx[0]].tobytes()]
    st = server.keyset_stats()
    assert st["resident"] == 2 and st["capacity"] == 2
    assert st["evictions"] == 4 and st["key_uploads"] == 6 * n_keys   # 3 clients round robin on 2 slots: always a miss
    # more clients in one ProcessRequests call than slots: served in windows of `capacity` clients
    wants = [[5], [6], [7]]
    out = server.ProcessRequests([c.CreateRequest(w) for c, w in zip(clients, wants)])
    for c, w, (st_, resp) in zip(clients, wants, out):
        assert st_ == 0 and c.ProcessResponse(w, resp) == [raw[w[0]].tobytes()]


def test_fingerprint_match_with_different_bytes_does_not_reuse_the_keys():
    """A lone single-query request starts on a match of length + sampled fingerprint and verifies the key bytes while
    the GPU works.  A key object that differs from a resident one only in bytes the fingerprint does not sample must
    be treated as another client's: the reply has to be the one a fresh server computes for it."""
    import seal_wire as W
    pp, raw, server, clients = _product_setup(n_clients=1)
    c = clients[0]
    enc = pp.encryption_parameters
    N, mods = enc.poly_modulus_degree, enc.coeff_modulus
    key_pid, data_pid = W.parms_id(N, mods, enc.plain_modulus), W.parms_id(N, mods[:-1], enc.plain_modulus)
    keys = c.galois_keys()
    query = c.create_query_for(123)
    blob_a = W.save_galois_keys(keys, N, key_pid)
    words = len(blob_a) // 8
    step = max(1, words // 64)                  # the library samples 8-byte words 0, step, 2 step, ... of the object
    blob_b = None
    for coeff in range(7, 200):                 # change one key coefficient until the changed word is not a sampled one
        k2 = {g: v.copy() for g, v in keys.items()}
        g0 = sorted(k2)[-1]                      # N + 1: the first expansion level always uses it
        k2[g0][1, 0, 1, coeff] ^= np.uint64(1)
        cand = W.save_galois_keys(k2, N, key_pid)
        diff = [i for i in range(0, len(cand), 8) if cand[i:i + 8] != blob_a[i:i + 8]]
        if len(cand) == len(blob_a) and diff and all((i // 8) % step for i in diff):
            blob_b = cand
            break
    assert blob_b is not None
    req_a, req_b = W.save_request([query], blob_a, data_pid), W.save_request([query], blob_b, data_pid)
    good = server.ProcessRequest(req_a)
    assert c.ProcessResponse([123], good) == [raw[123].tobytes()]
    got = server.ProcessRequest(req_b)          # same length, same fingerprint, different key bytes
    fresh = pir_amd.PIRServer.Create(pir_amd.PIRDatabase.Create(pp, raw), pp)
    want = fresh.ProcessRequest(req_b)
    assert got == want and got != good
    assert server.keyset_stats()["resident"] == 2            # the other object became its own key set
    assert server.ProcessRequest(req_a) == good              # and the first client's set is intact
    # the same inside a WINDOW of requests (key bytes compared on worker threads under the batch's GPU work): on a fresh
    # server client A's keys become resident first; one call then carries a two-query request with A's object, one with
    # the look-alike and another with A's -- the look-alike must come back as its own client's replies, A's unchanged
    server2 = pir_amd.PIRServer.Create(pir_amd.PIRDatabase.Create(pp, raw), pp)
    assert server2.ProcessRequest(req_a) == good
    req_a2 = W.save_request([query, c.create_query_for(77)], blob_a, data_pid)
    res = server2.ProcessRequests([req_a2, req_b, req_a])
    assert [st for st, _ in res] == [0, 0, 0]
    assert res[1][1] == want and res[2][1] == good
    assert c.ProcessResponse([123, 77], res[0][1]) == [raw[123].tobytes(), raw[77].tobytes()]
    assert server2.keyset_stats()["resident"] == 2
    # and with the look-alike FIRST in the window of a third server whose only resident set is A's
    server3 = pir_amd.PIRServer.Create(pir_amd.PIRDatabase.Create(pp, raw), pp)
    assert server3.ProcessRequest(req_a) == good
    res = server3.ProcessRequests([req_b, req_a, req_b])
    assert [st for st, _ in res] == [0, 0, 0] and res[0][1] == want and res[1][1] == good and res[2][1] == want


def test_many_threads_many_clients_are_combined_correctly():
    """8 threads hammer ONE context with requests of 4 clients (one or two queries each, some threads sharing a
    client): whichever thread leads serves the queued requests together; every response must decode under its own
    client's keys and equal what the same request gets when served alone."""
    pp, raw, server, clients = _product_setup(n_clients=4)
    errors, results = [], {}
    lock = threading.Lock()

    def work(t):
        try:
            c = clients[t % 4]
            for it in range(5):
                idx = [(131 * t + 17 * it) % 2000] + ([(7 * t + 3 * it + 1) % 2000] if (t + it) % 3 == 0 else [])
                req = c.CreateRequest(idx)
                resp = server.ProcessRequest(req)
                if c.ProcessResponse(idx, resp) != [raw[i].tobytes() for i in idx]:
                    errors.append(("decode", t, it))
                with lock:
                    results[(t, it)] = (req, resp)
        except Exception as ex:            # noqa: BLE001
            errors.append((t, repr(ex)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    assert len(results) == 40
    for (t, it), (req, resp) in sorted(results.items())[::7]:       # the server is deterministic: alone = combined
        assert server.ProcessRequest(req) == resp, (t, it)
    st = server.keyset_stats()
    assert st["resident"] == 4 and st["evictions"] == 0
    assert st["key_uploads"] == 4 * len(pir_amd.generate_galois_elts(4096))


def test_malformed_requests_inside_a_batch_fail_alone():
    """Truncations, bit flips and garbage between good requests of two clients: every bad request gets its own
    InvalidArgument, every good one the bytes it gets when served alone; nothing is left half-installed."""
    pp, raw, server, clients = _product_setup(n_clients=2)
    good = [clients[0].CreateRequest([5]), clients[1].CreateRequest([6, 7]), clients[0].CreateRequest([8])]
    alone = [server.ProcessRequest(r) for r in good]
    rng = np.random.default_rng(11)
    bad = [good[0][:len(good[0]) // 2], good[1][:-1], b"\x0a\x05hello", b"", bytes(rng.integers(0, 256, 300, dtype=np.uint8))]
    flipped = bytearray(good[2])
    flipped[40] ^= 0xFF                                   # inside the first ciphertext's SEAL header
    bad.append(bytes(flipped))
    mixed = [bad[0], good[0], bad[1], bad[2], good[1], bad[3], bad[4], good[2], bad[5]]
    out = server.ProcessRequests(mixed)
    expect_good = {1: alone[0], 4: alone[1], 7: alone[2]}
    for i, (st, resp) in enumerate(out):
        if i in expect_good:
            assert st == 0 and resp == expect_good[i], i
        elif mixed[i] == b"":
            # an empty Request has no keys: SEALDeserialize<GaloisKeys> of empty bytes throws -> InvalidArgument
            assert st == pir_amd.StatusCode.INVALID_ARGUMENT, i
        else:
            assert st == pir_amd.StatusCode.INVALID_ARGUMENT and resp is None, (i, st)
    assert server.keyset_stats()["resident"] == 2
    for r, a in zip(good, alone):                          # the context is still fine afterwards
        assert server.ProcessRequest(r) == a


def test_more_queries_than_one_batch_chunk():
    """3 clients x 30 queries = 90 queries in one pirgpu_process_requests call: served in chunks of <= 64 queries that
    cut through a request; reply i of every response still answers query i of its request (server.cpp:60-63)."""
    pp, raw, server, clients = _product_setup(items=600, elem=64, d=2, n_clients=3)
    wants = [[(17 * c + 7 * i) % 600 for i in range(30)] for c in range(3)]
    requests = [cl.CreateRequest(w) for cl, w in zip(clients, wants)]
    out = server.ProcessRequests(requests)
    for cl, w, (st, resp) in zip(clients, wants, out):
        assert st == 0
        assert cl.ProcessResponse(w, resp) == [raw[i].tobytes() for i in w]
    assert server.ProcessRequest(requests[1]) == out[1][1]


def test_stale_key_set_handles_are_refused_not_reused():
    """slots handed out by the direct API are HANDLES (slot index + generation).  Once a set has been
    evicted, its old handle fails with FailedPrecondition everywhere -- also where the context REMEMBERS it (a staged
    batch, the single-query selection) -- instead of silently switching a query with the new tenant's keys."""
    s = PirSetup(3000, 288, 2, N=4096, plain_bits=24)
    p = s.params
    db, srv = _server(s)
    srv.set_keyset_capacity(2)
    clients = [Client(s.orc, seed=4000 + i) for i in range(3)]
    keys = [c.galois_keys() for c in clients]
    a = srv.install_keyset(b"client-a", keys[0])
    b = srv.install_keyset(b"client-b", keys[1])
    queries = np.stack([clients[i].create_query_for(p, 100 + i) for i in range(2)])
    srv.set_concurrency(8)
    srv.stage_batch(queries)
    srv.set_batch_keysets([a, b])
    srv.run_batch()
    got = srv.fetch_batch()
    for i in range(2):
        rc, exp = s.orc.process_query(s.db_ntt, p.dimensions, queries[i], keys[i])
        assert rc == 0 and np.array_equal(got[i], exp)
    # a third client pushes the least recently used set (client a's) out while the batch that names it is still staged
    c = srv.install_keyset(b"client-c", keys[2])
    assert srv.keyset_stats()["evictions"] == 1
    assert (c & 0xFFF) == (a & 0xFFF) and c != a                 # same slot, next generation
    with pytest.raises(pir_amd.PirGpuError) as e:                # the staged batch remembers a's handle: refused, not
        srv.run_batch()                                          # run with client c's keys
    assert e.value.code == pir_amd.StatusCode.FAILED_PRECONDITION and "stale" in e.value.message
    for stale_use in (lambda: srv.use_keyset(a),
                      lambda: (srv.stage_batch(queries), srv.set_batch_keysets([a, b])),
                      lambda: srv.release_keyset(a)):
        with pytest.raises(pir_amd.PirGpuError) as e:
            stale_use()
        assert e.value.code == pir_amd.StatusCode.FAILED_PRECONDITION and "stale" in e.value.message
    # the live handles keep working: client c in the slot client a had, client b untouched
    q2 = np.stack([clients[2].create_query_for(p, 777), clients[1].create_query_for(p, 778)])
    srv.stage_batch(q2)
    srv.set_batch_keysets([c, b])
    srv.run_batch()
    got = srv.fetch_batch()
    for qi, ki in ((0, 2), (1, 1)):
        rc, exp = s.orc.process_query(s.db_ntt, p.dimensions, q2[qi], keys[ki])
        assert rc == 0 and np.array_equal(got[qi], exp)
    # the single-query selection is checked the same way
    srv.use_keyset(c)
    srv.use_keyset(b)                                            # b selected; c is now the least recently used
    d2 = srv.install_keyset(b"client-a-again", keys[0])          # evicts c
    assert (d2 & 0xFFF) == (c & 0xFFF)
    assert np.array_equal(srv.process_query(q2[1]), got[1])      # b: still there
    srv.use_keyset(d2)
    e2 = srv.install_keyset(b"client-c-again", keys[2])          # evicts b (d2 was used last)
    assert (e2 & 0xFFF) == (b & 0xFFF)
    e3 = srv.install_keyset(b"client-b-again", keys[1])          # evicts d2 -- the selected set
    assert (e3 & 0xFFF) == (d2 & 0xFFF)
    with pytest.raises(pir_amd.PirGpuError) as e:
        srv.process_query(q2[1])
    assert e.value.code == pir_amd.StatusCode.FAILED_PRECONDITION and "stale" in e.value.message
    srv.use_keyset(0)
    db.close()
