"""GPU parity tests: the HIP path (through the C ABI of libpirgpu.so) against the CPU
oracle on the same seeded inputs -- bit-exact on every residue -- and against the
reference's plaintext-level known answers through the oracle's CPU client."""
import numpy as np
import pytest

import oracle
import pir_amd
from conftest import parse_poly
from gpu_helpers import random_ct, random_key, to_product_params
from pir_fixtures import PirSetup

pytestmark = pytest.mark.gpu

N = 4096


def make_server(setup: PirSetup, load_db=True, shard=None):
    pp = to_product_params(setup.params)
    db = pir_amd.PIRDatabase.Create(pp, shard=shard)
    if load_db:
        db.populate(setup.raw)
    srv = pir_amd.PIRServer(db, pp) if shard is not None else pir_amd.PIRServer.Create(db, pp) if load_db \
        else pir_amd.PIRServer(db, pp)
    return db, srv


@pytest.fixture(scope="module")
def small():
    # server_test.cpp:60-84 fixture: N=4096, 20-bit t, 10-item DB
    s = PirSetup(10, 0, 1, N=N, plain_bits=20)
    db, srv = make_server(s)
    srv.set_galois_keys(s.galois_keys)
    return s, db, srv


# ---------------------------------------------------------------- NTT (K4/K7)

@pytest.mark.parametrize("Nn,bits", [(4096, 20), (8192, 20), (2048, 14), (16384, 24)])
def test_ntt_parity(Nn, bits):
    if Nn == 2048:
        moduli = oracle.coeff_modulus_create(2048, [27, 27])
    elif Nn == 8192:
        m = oracle.BFV_DEFAULT[8192]
        moduli = m[:3] + [m[4]]
    elif Nn == 16384:
        m = oracle.BFV_DEFAULT[16384]
        moduli = m[:4] + [m[8]]
    else:
        moduli = oracle.BFV_DEFAULT[Nn]
    s = PirSetup(4, 0, 1, N=Nn, plain_bits=bits, moduli=moduli)
    db, srv = make_server(s, load_db=False)
    rng = np.random.default_rng(Nn)
    cts = random_ct(s.orc, rng, 3)
    fwd = srv.ntt_forward(cts)
    exp = np.stack([s.orc.ct_ntt_fwd(c) for c in cts])
    assert np.array_equal(fwd, exp)
    assert np.array_equal(srv.ntt_inverse(fwd), cts)
    # key level: [k+1][N] over q_0..q_{k-1}, p
    kl = np.empty((2, s.orc.k + 1, Nn), dtype=np.uint64)
    for i in range(s.orc.k + 1):
        kl[:, i, :] = rng.integers(0, s.orc.moduli[i], size=(2, Nn), dtype=np.uint64)
    fk = srv.ntt_forward(kl, key_level=True)
    for b in range(2):
        for i in range(s.orc.k + 1):
            assert np.array_equal(fk[b, i], s.orc.ntt_fwd(i, kl[b, i]))
    assert np.array_equal(srv.ntt_inverse(fk, key_level=True), kl)
    db.close()


# ---------------------------------------------------------------- database encode (K5)

@pytest.mark.parametrize("dbsize,elem,bpc,pbits", [(10, 0, 0, 20), (1200, 64, 10, 24), (1200, 289, 10, 24),
                                                   (300, 288, 0, 24), (77, 100, 7, 16)])
def test_db_encode_parity(dbsize, elem, bpc, pbits):
    s = PirSetup(dbsize, elem, 1, N=N, plain_bits=pbits, bits_per_coeff=bpc)
    db, srv = make_server(s)
    assert db.size() == s.params.num_pt
    for i in range(s.params.num_pt):
        assert np.array_equal(db.read_plaintext(i), s.db_ntt[i]), i
    db.close()


def test_db_load_coeffs_parity():
    s = PirSetup(10, 0, 1, N=N, plain_bits=20)
    pp = to_product_params(s.params)
    db = pir_amd.PIRDatabase.Create(pp)
    rng = np.random.default_rng(2)
    rows = [rng.integers(0, s.orc.t, size=rng.integers(1, N + 1), dtype=np.uint64) for _ in range(10)]
    db.populate_coeffs(rows)
    exp = s.orc.db_from_coeffs(rows)
    for i in range(10):
        assert np.array_equal(db.read_plaintext(i), exp[i])
    db.close()


def test_db_size_mismatch_is_invalid_argument():
    s = PirSetup(10, 64, 1, N=N, plain_bits=20)
    pp = to_product_params(s.params)
    db = pir_amd.PIRDatabase.Create(pp)
    with pytest.raises(pir_amd.PirGpuError) as e:
        db.populate(s.raw[:9])
    assert e.value.code == pir_amd.StatusCode.INVALID_ARGUMENT     # database.cpp:85-90
    db2 = pir_amd.PIRDatabase.Create(pp)
    with pytest.raises(pir_amd.PirGpuError) as e:                  # server.cpp:37-39
        pir_amd.PIRServer.Create(db2, pp)
    assert e.value.code == pir_amd.StatusCode.INVALID_ARGUMENT


def test_ct_multiplication_mode_unimplemented():
    s = PirSetup(10, 0, 1, N=N, plain_bits=20)
    pp = to_product_params(s.params)
    pp.use_ciphertext_multiplication = True
    with pytest.raises(pir_amd.PirGpuError) as e:
        pir_amd.PIRDatabase.Create(pp)
    assert e.value.code == pir_amd.StatusCode.UNIMPLEMENTED


# ---------------------------------------------------------------- substitution (K1)

@pytest.mark.parametrize("g", [3, 5, N + 1, N // 2 + 1, N // 8 + 1, 2 * N - 1, 9])
def test_substitute_parity_random(small, g):
    s, db, srv = small
    rng = np.random.default_rng(g)
    ct = random_ct(s.orc, rng)[0]
    key = random_key(s.orc, rng)
    srv.set_galois_keys({g: key})
    rc, exp = s.orc.apply_galois_ct(ct, g, key)
    assert rc == 0
    got = srv.substitute_power_x_inplace(ct.copy(), g)
    assert np.array_equal(got, exp)
    srv.set_galois_keys(s.galois_keys)


SUBSTITUTIONS = [("42", 3, "42"), ("1x^1", 5, "1x^5"), ("6x^2", 3, "6x^6"), ("1x^1", N + 1, "FC000x^1"),
                 ("1x^8", N // 8 + 1, "FC000x^8"), ("77x^4095", 3, "77x^4093"),
                 ("4x^4 + 33x^3 + 222x^2 + 19x^1 + 42", N + 1, "4x^4 + FBFCEx^3 + 222x^2 + FBFE8x^1 + 42")]


@pytest.mark.parametrize("inp,power,expected", SUBSTITUTIONS)
def test_substitute_known_answers(small, inp, power, expected):
    # server_test.cpp:291-305
    s, db, srv = small
    srv.set_galois_keys({power: s.client.galois_key(power)})
    ct = s.client.encrypt(parse_poly(inp, N))
    out = srv.substitute_power_x_inplace(ct.copy(), power)
    assert (s.client.decrypt(out) == parse_poly(expected, N)).all()
    srv.set_galois_keys(s.galois_keys)


def test_substitute_missing_key_is_internal(small):
    s, db, srv = small
    srv.set_galois_keys({})
    with pytest.raises(pir_amd.PirGpuError) as e:
        srv.substitute_power_x_inplace(random_ct(s.orc, np.random.default_rng(0))[0], 3)
    assert e.value.code == pir_amd.StatusCode.INTERNAL      # server.cpp:72-74
    srv.set_galois_keys(s.galois_keys)


# ---------------------------------------------------------------- x^-k (K2)

@pytest.mark.parametrize("k", [0, 1, 4, 41, N - 1, N, N + 4, 2 * N - 1])
def test_multiply_inverse_power_of_x_parity(small, k):
    s, db, srv = small
    ct = random_ct(s.orc, np.random.default_rng(k))[0]
    assert np.array_equal(srv.multiply_inverse_power_of_x(ct, k), s.orc.multiply_inverse_power_of_x(ct, k))


def test_multiply_inverse_power_of_x_known_answers(small):
    # server_test.cpp:333-339
    s, db, srv = small
    for inp, k, expected in [("42x^1", 1, "42"), ("42x^42", 41, "42x^1"), ("1x^4 + 1x^3 + 1x^1", 1, "1x^3 + 1x^2 + 1"),
                             ("1x^16 + 1x^12 + 1x^8", 4, "1x^12 + 1x^8 + 1x^4")]:
        out = srv.multiply_inverse_power_of_x(s.client.encrypt(parse_poly(inp, N)), k)
        assert (s.client.decrypt(out) == parse_poly(expected, N)).all()


# ---------------------------------------------------------------- expansion

@pytest.mark.parametrize("n", [1, 2, 3, 4, 6, 10, 16])
def test_expansion_parity(small, n):
    s, db, srv = small
    ct = s.client.encrypt(parse_poly("3x^3 + 2x^2 + 1x^1 + 42", N))
    rc, exp = s.orc.oblivious_expansion(ct, n, s.galois_keys)
    assert rc == 0
    got = srv.oblivious_expansion(ct, n)
    assert np.array_equal(got, exp)


def test_expansion_known_answers(small):
    # server_test.cpp:376-383
    s, db, srv = small
    for inp, expected in [("1", ["2", "0"]), ("1x^1", ["0", "2"]), ("3x^3 + 2x^2 + 1x^1 + 42", ["108", "4", "8", "C"]),
                          ("1x^5", ["0", "0", "0", "0", "0", "8"])]:
        res = srv.oblivious_expansion(s.client.encrypt(parse_poly(inp, N)), len(expected))
        for ct, e in zip(res, expected):
            assert (s.client.decrypt(ct) == parse_poly(e, N)).all()


def test_expansion_errors(small):
    s, db, srv = small
    ct = s.client.encrypt(parse_poly("1", N))
    with pytest.raises(pir_amd.PirGpuError) as e:
        srv.oblivious_expansion(ct, N + 1)
    assert e.value.code == pir_amd.StatusCode.INVALID_ARGUMENT      # server.cpp:111-114
    with pytest.raises(pir_amd.PirGpuError) as e:
        srv.oblivious_expansion(np.stack([ct]), N)                   # needs total/N + 1 cts (server.cpp:154-158)
    assert e.value.code == pir_amd.StatusCode.INVALID_ARGUMENT


def test_expansion_full_tree_and_multi_ct():
    # server_test.cpp:423-428: (5000 items, index 4200) -> 1024, (4096, 3007) -> 4096
    s = PirSetup(5000, 0, 1, N=N, plain_bits=20)
    db, srv = make_server(s, load_db=False)
    srv.set_galois_keys(s.galois_keys)
    for num_items, index, value in [(5000, 4200, 1024), (5000, 4095, 4096), (100, 42, 128)]:
        cts = []
        for i in range(num_items // N + 1):
            pt = np.zeros(N, dtype=np.uint64)
            if index // N == i:
                pt[index % N] = 1
            cts.append(s.client.encrypt(pt))
        res = srv.oblivious_expansion(np.stack(cts), num_items)
        assert res.shape[0] == num_items
        for i in sorted({0, 1, index - 1, index, index + 1, num_items - 1, 4095, 4096} & set(range(num_items))):
            d = s.client.decrypt(res[i])
            assert int(d[0]) == (value if i == index else 0) and not d[1:].any(), i
    # bit-exact against the oracle on one full 4096-leaf tree is covered by spot-checking leaves
    ct = s.client.encrypt(parse_poly("1x^77", N))
    rc, exp = s.orc.oblivious_expansion(ct, 64, s.galois_keys)
    assert np.array_equal(srv.oblivious_expansion(ct, 64), exp)
    db.close()


# ---------------------------------------------------------------- multiply / full query

@pytest.mark.parametrize("dbsize,elem,d,bpc", [(10, 0, 1, 0), (9, 0, 2, 10), (500, 0, 2, 6), (1200, 64, 1, 10),
                                               (256, 288, 2, 0), (27, 0, 3, 0), (2000, 288, 1, 0)])
def test_multiply_and_query_parity(dbsize, elem, d, bpc):
    s = PirSetup(dbsize, elem, d, N=N, plain_bits=24, bits_per_coeff=bpc)
    db, srv = make_server(s)
    srv.set_galois_keys(s.galois_keys)
    idx = dbsize // 2 + 1
    q = s.client.create_query_for(s.params, idx)
    # PIRDatabase::multiply on the oracle-expanded selection vector
    rc, sv = s.orc.oblivious_expansion_multi(q, s.params.dim_sum, s.galois_keys)
    assert rc == 0
    rc, exp = s.orc.db_multiply(s.db_ntt, s.params.dimensions, sv.copy())
    assert rc == 0
    assert np.array_equal(db.multiply(sv), exp)
    # whole processQuery
    got = srv.process_query(q)
    assert np.array_equal(got, exp)
    assert s.client.process_response(s.params, idx, got) == s.item(idx)
    db.close()


def test_multiply_selection_vector_size_mismatch(small):
    s, db, srv = small
    with pytest.raises(pir_amd.PirGpuError) as e:
        db.multiply(s.orc.new_ct(9))
    assert e.value.code == pir_amd.StatusCode.INVALID_ARGUMENT      # database.cpp:297-300


def test_query_wrong_ct_count(small):
    s, db, srv = small
    with pytest.raises(pir_amd.PirGpuError) as e:
        srv.process_query(s.orc.new_ct(2))
    assert e.value.code == pir_amd.StatusCode.INVALID_ARGUMENT


@pytest.mark.parametrize("dbsize,elem,d", [(300, 288, 2), (1500, 288, 1), (64, 0, 3)])
def test_sharded_partial_replies_sum_to_full_reply(dbsize, elem, d):
    """Row shards (as each GPU of a node would hold) give partial replies whose mod-q sum is the reply."""
    s = PirSetup(dbsize, elem, d, N=N, plain_bits=24)
    q = s.client.create_query_for(s.params, dbsize - 3)
    rc, exp = s.orc.process_query(s.db_ntt, s.params.dimensions, q, s.galois_keys)
    n0 = s.params.dimensions[0]
    cuts = [0, n0 // 3, n0 // 3, (2 * n0) // 3 + 1, n0]     # includes an empty shard
    acc = np.zeros(exp.shape, dtype=np.uint64)
    for a, b in zip(cuts[:-1], cuts[1:]):
        if a == b and a == 0:
            continue
        db, srv = make_server(s, shard=(a, b))
        srv.set_galois_keys(s.galois_keys)
        acc += srv.process_query(q)
        db.close()
    for j, qj in enumerate(s.orc.moduli[: s.orc.k]):
        acc[:, :, j, :] %= np.uint64(qj)
    assert np.array_equal(acc, exp)


@pytest.mark.parametrize("n_workers,count", [(1, 3), (3, 5), (4, 4)])
def test_batch_mode_matches_single_queries(n_workers, count):
    """Several queries in flight on separate workers give exactly the single-query replies, in order."""
    s = PirSetup(300, 288, 2, N=N, plain_bits=24)
    db, srv = make_server(s)
    srv.set_galois_keys(s.galois_keys)
    indexes = [(37 * i + 5) % 300 for i in range(count)]
    queries = np.stack([s.client.create_query_for(s.params, i) for i in indexes])
    got = srv.process_batch(queries, n_workers=n_workers)
    assert got.shape[0] == count
    for i, idx in enumerate(indexes):
        rc, exp = s.orc.process_query(s.db_ntt, s.params.dimensions, queries[i], s.galois_keys)
        assert np.array_equal(got[i], exp)
        assert s.client.process_response(s.params, idx, got[i]) == s.item(idx)
    # the single-query path still works afterwards on worker 0
    assert np.array_equal(srv.process_query(queries[0]), got[0])
    with pytest.raises(pir_amd.PirGpuError) as e:
        srv.stage_batch(np.concatenate([queries, queries], axis=1))     # wrong ciphertext count per query
    assert e.value.code == pir_amd.StatusCode.INVALID_ARGUMENT
    db.close()


# ---------------------------------------------------------------- wire level (ProcessRequest)

def test_process_request_wire_roundtrip():
    """server_test.cpp:98-186 shape: serialized pir.Request (payload.proto) -> pir.Response, single,
    batch and 2-dim, against the residue-level path."""
    import seal_wire as W
    s = PirSetup(82, 0, 2, N=N, plain_bits=24)
    db, srv = make_server(s)
    o = s.orc
    data_pid = W.parms_id(N, o.moduli[: o.k], o.t)
    key_pid = W.parms_id(N, o.moduli, o.t)
    gk = W.save_galois_keys(s.galois_keys, N, key_pid)
    indexes = [3, 42, 81]
    queries = [s.client.create_query_for(s.params, i) for i in indexes]
    resp = srv.ProcessRequest(W.save_request(queries, gk, data_pid))
    replies = W.load_response(resp)
    assert len(replies) == len(indexes)                       # reply[i] answers query[i] (server.cpp:60-63)
    srv.set_galois_keys(s.galois_keys)
    for idx, q, r in zip(indexes, queries, replies):
        assert np.array_equal(r, srv.process_query(q))
        assert s.client.process_response(s.params, idx, r) == s.item(idx)
    # zero queries -> empty response
    assert srv.ProcessRequest(W.save_request([], gk, data_pid)) == b""
    db.close()


def test_process_request_wire_errors():
    import seal_wire as W
    s = PirSetup(10, 0, 1, N=N, plain_bits=20)
    db, srv = make_server(s)
    o = s.orc
    data_pid = W.parms_id(N, o.moduli[: o.k], o.t)
    key_pid = W.parms_id(N, o.moduli, o.t)
    gk = W.save_galois_keys(s.galois_keys, N, key_pid)
    q = s.client.create_query_for(s.params, 3)
    for bad in (W.save_request([q], b"", data_pid),                      # empty galois_keys: load throws
                W.save_request([q], gk[:100], data_pid),                 # truncated keys
                W.save_request([q], gk, key_pid),                        # query at the wrong level
                W.save_request([np.concatenate([q, q])], gk, data_pid),  # wrong ciphertext count
                b"\x0a\xff\xff"):                                        # malformed proto
        with pytest.raises(pir_amd.PirGpuError) as e:
            srv.ProcessRequest(bad)
        assert e.value.code == pir_amd.StatusCode.INVALID_ARGUMENT       # serialization.h:113-115
    # a coefficient >= q_j is rejected like SEAL's is_data_valid_for
    q_bad = q.copy()
    q_bad[0, 0, 0, 0] = o.moduli[0]
    with pytest.raises(pir_amd.PirGpuError) as e:
        srv.ProcessRequest(W.save_request([q_bad], gk, data_pid))
    assert e.value.code == pir_amd.StatusCode.INVALID_ARGUMENT
    db.close()


# ---------------------------------------------------------------- BASELINE.json parameter sets

def _baseline_moduli(cfg):
    """Moduli choices for the BASELINE.json configs."""
    if cfg == 1:   # N=2048, 1 data prime + special, total <= 54 bits: CoeffModulus::Create(2048, {27, 27})
        return 2048, oracle.coeff_modulus_create(2048, [27, 27]), oracle.plain_modulus_batching(2048, 14)
    if cfg == 4:   # N=8192, 3 data primes: first 3 of BFVDefault(8192) + its last prime as the special prime
        m = oracle.BFV_DEFAULT[8192]
        return 8192, m[:3] + [m[4]], oracle.plain_modulus_batching(8192, 24)
    if cfg == 5:   # N=16384, 4 data primes: first 4 of BFVDefault(16384) + its last prime as the special prime
        m = oracle.BFV_DEFAULT[16384]
        return 16384, m[:4] + [m[8]], oracle.plain_modulus_batching(16384, 24)
    raise ValueError(cfg)


@pytest.mark.parametrize("cfg,dbsize,elem,d", [(1, 1 << 10, 32, 1), (4, 600, 1024, 2), (5, 2000, 288, 2)])
def test_baseline_parameter_sets(cfg, dbsize, elem, d):
    """configs[0] exactly, configs[3]/[4] with their ring / modulus chain at a reduced item count:
    GPU reply bit-exact vs the oracle and the item decodes."""
    Nn, moduli, t = _baseline_moduli(cfg)
    s = PirSetup(dbsize, elem, d, N=Nn, moduli=moduli, t=t)
    if cfg == 1:
        assert s.params.items_per_plaintext == 104 and s.params.num_pt == 10 and s.orc.expansion_ratio() == 3
    if cfg == 4:
        assert s.params.items_per_plaintext == 23 and s.orc.expansion_ratio() == 6
    if cfg == 5:
        assert s.params.items_per_plaintext == 163 and s.orc.expansion_ratio() == 12
    db, srv = make_server(s)
    srv.set_galois_keys(s.galois_keys)
    assert db.reply_ct_count() == s.orc.reply_ct_count(d)
    idx = dbsize - 7
    q = s.client.create_query_for(s.params, idx)
    rc, exp = s.orc.process_query(s.db_ntt, s.params.dimensions, q, s.galois_keys)
    assert rc == 0
    got = srv.process_query(q)
    assert np.array_equal(got, exp)
    if cfg != 1:   # configs[0] has a thin noise margin by construction; parity is what is pinned
        assert s.client.process_response(s.params, idx, got) == s.item(idx)
    db.close()


def test_multi_ciphertext_query_end_to_end():
    """dim_sum > N: the query is two ciphertexts (server_test.cpp:124-151, client.cpp:109-134)."""
    s = PirSetup(4200, 0, 1, N=N, plain_bits=20)
    assert s.params.dim_sum == 4200 and s.params.num_pt == 4200
    db, srv = make_server(s)
    srv.set_galois_keys(s.galois_keys)
    idx = 4150
    q = s.client.create_query_for(s.params, idx)
    assert q.shape[0] == 2
    rc, exp = s.orc.process_query(s.db_ntt, s.params.dimensions, q, s.galois_keys)
    assert rc == 0
    got = srv.process_query(q)
    assert np.array_equal(got, exp)
    assert s.client.process_response(s.params, idx, got) == s.item(idx)
    db.close()


def test_batch_mode_one_dimension():
    """d=1 (column-split scan, no shared pass) through the batch API."""
    s = PirSetup(1500, 288, 1, N=N, plain_bits=24)
    db, srv = make_server(s)
    srv.set_galois_keys(s.galois_keys)
    indexes = [0, 749, 1499]
    queries = np.stack([s.client.create_query_for(s.params, i) for i in indexes])
    got = srv.process_batch(queries, n_workers=2)
    for i, idx in enumerate(indexes):
        rc, exp = s.orc.process_query(s.db_ntt, s.params.dimensions, queries[i], s.galois_keys)
        assert np.array_equal(got[i], exp)
        assert s.client.process_response(s.params, idx, got[i]) == s.item(idx)
    db.close()


@pytest.mark.parametrize("world,count", [(2, 4), (4, 4)])
def test_query_parallel_expansion_simulated_ranks(world, count):
    """The multi-GPU step (pir_amd.distributed.run_batch_query_parallel) simulated on one GPU: each
    'rank' = a context holding one row shard expands only its block of the batch into a shared
    device buffer (what the RCCL all-gather assembles), then every rank multiplies ALL queries
    against its shard; the mod-q sum of the partial replies must equal the full replies."""
    import torch
    from pir_amd.distributed import owned_queries, shard_range
    s = PirSetup(300, 288, 2, N=N, plain_bits=24)
    p = s.params
    indexes = [(53 * i + 11) % 300 for i in range(count)]
    queries = np.stack([s.client.create_query_for(p, i) for i in indexes])
    sv_all = torch.zeros((count, p.dim_sum, 2, s.orc.k, N), dtype=torch.int64, device="cuda")
    ranks = []
    for r in range(world):
        db, srv = make_server(s, shard=shard_range(p.dimensions[0], r, world))
        srv.set_galois_keys(s.galois_keys)
        srv.set_concurrency(2)
        srv.stage_batch(queries)
        lo, hi = owned_queries(count, r, world)
        srv.batch_expand(lo, hi - lo, sv_all[lo].data_ptr())
        ranks.append((db, srv))
    torch.cuda.synchronize()
    # the gathered selection vectors are exactly the oracle's expansion in NTT form
    rc, sv0 = s.orc.oblivious_expansion_multi(queries[0], p.dim_sum, s.galois_keys)
    got0 = ranks[0][1].ntt_inverse(sv_all[0].cpu().numpy().view(np.uint64))   # SEAL order in -> coefficients
    # (sv_all is in device NTT order, so compare through the full reply instead of this hook)
    acc = None
    for db, srv in ranks:
        srv.batch_run_selectors(sv_all.data_ptr(), count)
        part = srv.fetch_batch()
        acc = part.copy() if acc is None else acc + part
    for j, qj in enumerate(s.orc.moduli[: s.orc.k]):
        acc[:, :, :, j, :] %= np.uint64(qj)
    for i, idx in enumerate(indexes):
        rc, exp = s.orc.process_query(s.db_ntt, p.dimensions, queries[i], s.galois_keys)
        assert rc == 0 and np.array_equal(acc[i], exp), i
        assert s.client.process_response(p, idx, acc[i]) == s.item(idx)
    for db, srv in ranks:
        db.close()


def test_full_size_benchmark_workload():
    """BASELINE.json configs[2] at FULL size (N=4096, 2^20 x 288 B, d=2, dims 162x162).

    With the reference's benchmark parameters the reply's noise budget is already ~0.4 bit at 2^16
    items (the largest size benchmark.cpp sweeps) and negative at 2^20, so decrypting the reply is not
    a usable property here; what is checked at full size is (1) the GPU reply equals the CPU oracle's
    bit for bit for a real client query, (2) the batch API returns the same replies as single queries,
    (3) row shards' partial replies sum (mod q) to the unsharded reply bit for bit."""
    rng = np.random.default_rng(2026)
    n_items = 1 << 20
    params = oracle.create_pir_parameters(n_items, 288, 2, N=N, plain_bits=24)
    assert params.dimensions == [162, 162] and params.num_pt == 26215
    orc = oracle.Oracle.from_params(params)
    from oracle.client import Client
    client = Client(orc, seed=5)
    keys = client.galois_keys()
    raw = rng.integers(0, 256, size=(n_items, 288), dtype=np.uint8)
    pp = to_product_params(params)
    db = pir_amd.PIRDatabase.Create(pp, raw)
    srv = pir_amd.PIRServer.Create(db, pp)
    srv.set_galois_keys(keys)
    indexes = [0, 524289, n_items - 1, 26214 * 40 + 3]
    queries = np.stack([client.create_query_for(params, i) for i in indexes])
    replies = srv.process_batch(queries, n_workers=4)
    for i in range(len(indexes)):
        assert np.array_equal(srv.process_query(queries[i]), replies[i]), i
    db.close()
    rc, db_ntt = orc.db_encode(raw.tobytes(), n_items, 288, params.items_per_plaintext, params.eff_bits_per_coeff,
                               params.num_pt)
    assert rc == 0
    rc, exp = orc.process_query(db_ntt, params.dimensions, queries[3], keys)
    assert rc == 0 and np.array_equal(replies[3], exp)
    del db_ntt
    acc = np.zeros_like(replies[1])
    for lo, hi in [(0, 50), (50, 161), (161, 162)]:
        dbs = pir_amd.PIRDatabase.Create(pp, raw, shard=(lo, hi))
        ss = pir_amd.PIRServer(dbs, pp)
        ss.set_galois_keys(keys)
        acc += ss.process_query(queries[1])
        dbs.close()
    for j, qj in enumerate(orc.moduli[: orc.k]):
        acc[:, :, j, :] %= np.uint64(qj)
    assert np.array_equal(acc, replies[1])
    # (4) SLOT shards at the headline shape: eight contexts, each holding 1 / 8 of the NTT slots of
    # every plaintext -- what the 8 ranks of `bench.py --gpus 8` hold --, sixteen queries (two per "rank"), the two
    # all-to-alls as tensor copies; the replies of the four queries above must be the same bits (one of them was checked
    # against the oracle's full pass), the rest must equal the plain pipeline's
    import torch
    from pir_amd import distributed as D
    from gpu_helpers import all_to_all_in_process
    G, per = 8, 2
    more = np.stack([client.create_query_for(params, (65537 * i + 11) % n_items) for i in range(G * per - len(indexes))])
    q16 = np.concatenate([queries, more])
    db = pir_amd.PIRDatabase.Create(pp, raw)
    srv = pir_amd.PIRServer.Create(db, pp)
    srv.set_galois_keys(keys)
    plain16 = srv.process_batch(q16, n_workers=16)
    assert np.array_equal(plain16[:4], replies)
    db.close()
    cuts = D.slot_cuts(orc.k * N, G)
    ranks = []
    for g in range(G):
        dbg = pir_amd.PIRDatabase.Create(pp, raw, slots=(cuts[g], cuts[g + 1]))
        dbg.finalize(release_staging=True)
        sg = pir_amd.PIRServer(dbg, pp)
        sg.set_galois_keys(keys)
        sg.set_concurrency(16)
        sg.stage_batch(q16)
        assert sg.scan_bytes() * G == 1141899264            # 1 / 8 of the packed database each
        ranks.append((dbg, sg, D.SlotsBuffers(sg, G * per, g, G, torch, "cuda:0")))
    bufs = [r[2] for r in ranks]
    for g, (_, sg, b) in enumerate(ranks):
        sg.slots_expand_async(g * per, per, b.packed_send.data_ptr(), b.sv.data_ptr(), cuts)
        sg.sync()
    all_to_all_in_process([b.packed_recv for b in bufs], [b.packed_send for b in bufs], [b.x1_recv for b in bufs],
                          [b.x1_send for b in bufs])
    for _, sg, b in ranks:
        sg.slots_scan_async(b.packed_recv.data_ptr(), G, per, b.rows_send.data_ptr())
        sg.sync()
    all_to_all_in_process([b.rows_recv for b in bufs], [b.rows_send for b in bufs], [b.x2_recv for b in bufs],
                          [b.x2_send for b in bufs])
    for g, (dbg, sg, b) in enumerate(ranks):
        sg.slots_finish_async(b.rows_recv.data_ptr(), per, b.sv.data_ptr(), cuts, b.replies.data_ptr())
        sg.sync()
        got = b.replies.cpu().numpy().view(np.uint64)
        for i in range(per):
            assert np.array_equal(got[i], plain16[g * per + i]), (g, i)
        dbg.close()
