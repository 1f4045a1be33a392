"""CPU tests of the product-side PIR client (libpirclient.so, include/pirclient.h): the query-layout and
response-decode tables of the reference's client_test.cpp, run against pir_amd.PIRClient, plus a
round trip through the CPU oracle server (an independent implementation of the server path), which
pins the NTT layout / key format the client emits to what the server side expects."""
import os
import re

import numpy as np
import pytest

import oracle
import pir_amd
from pir_amd import capi, parameters as P

import seal_wire
from gpu_helpers import to_product_params
from pir_fixtures import generate_test_db

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 4096     # client_test.cpp:36 POLY_MODULUS_DEGREE


def make_client(dbsize, dimensions=1, elem_size=0, use_ct_mult=False, seed=b"client-test"):
    """PIRClientTest::SetUpDB (client_test.cpp:42-52)."""
    enc = P.generate_encryption_params(N, 16)
    pp = P.create_pir_parameters(dbsize, elem_size, dimensions, enc, use_ct_mult)
    return pir_amd.PIRClient.Create(pp, seed=seed), pp


def data_pid(pp):
    e = pp.encryption_parameters
    return seal_wire.parms_id(e.poly_modulus_degree, e.coeff_modulus[:-1], e.plain_modulus)


def split_request(request: bytes):
    fields = seal_wire._parse(request)
    queries = [[seal_wire.load_ciphertext(bytes(p))[2] for n2, p in seal_wire._parse(payload) if n2 == 1]
               for num, payload in fields if num == 1]
    galois = [bytes(p) for num, p in fields if num == 2]
    relin = [bytes(p) for num, p in fields if num == 3]
    return queries, galois, relin


def test_client_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "pirclient.h")).read()
    declared = set(re.findall(r"\b(pirclient_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 15
    lib = capi.load_client()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(capi.CLIENT_SIGNATURES), declared ^ set(capi.CLIENT_SIGNATURES)


def test_encrypt_decrypt_roundtrip_and_noise_budget():
    c, pp = make_client(100)
    t = pp.encryption_parameters.plain_modulus
    rng = np.random.default_rng(1)
    pt = rng.integers(0, t, size=N, dtype=np.uint64)
    ct = c.encrypt(pt)
    assert (c.decrypt(ct) == pt).all()
    # fresh BFV noise at (N=4096, 72-bit Q, 16-bit t): SEAL reports about 50 bits
    assert 40 <= c.noise_budget(ct) <= 56
    short = c.encrypt([5, 0, t - 1])
    assert c.decrypt(short).tolist()[:4] == [5, 0, t - 1, 0]
    with pytest.raises(pir_amd.PirGpuError) as e:
        c.encrypt([t])
    assert e.value.code == pir_amd.StatusCode.INVALID_ARGUMENT


def test_seeded_clients_are_deterministic_and_unseeded_differ():
    a, _ = make_client(100, seed=b"s1")
    b, _ = make_client(100, seed=b"s1")
    d, _ = make_client(100, seed=None)
    assert (a.galois_key(N + 1) == b.galois_key(N + 1)).all()
    assert (a.create_query_for(7) == b.create_query_for(7)).all()
    assert not (a.galois_key(N + 1) == d.galois_key(N + 1)).all()


# ---------------------------------------------------------------- query layout (client_test.cpp:67-348)

def check_one_hot(c, pp, ct, expected, m):
    """decrypt(ct)[i] * m == 1 (mod t) at the expected slots, 0 elsewhere."""
    t = pp.encryption_parameters.plain_modulus
    pt = c.decrypt(ct)
    nz = set(np.nonzero(pt)[0].tolist())
    assert nz == set(expected)
    for i in expected:
        assert int(pt[i]) * m % t == 1


def test_create_request_d1():
    # client_test.cpp:67-93: db 100, index 5 -> slot 5 holds next_power_two(100)^-1
    c, pp = make_client(100)
    queries, galois, relin = split_request(c.CreateRequest([5]))
    assert len(queries) == 1 and len(queries[0]) == 1
    assert len(galois) == 1 and galois[0] and len(relin) == 1 and relin[0]
    check_one_hot(c, pp, queries[0][0], [5], P.next_power_two(100))


def test_create_request_d2():
    # client_test.cpp:95-127: 82 items, d=2 -> dims [10, 9]; index 42 -> row 4, col 6
    c, pp = make_client(82, 2)
    assert pp.dimensions == [10, 9]
    q = c.create_query_for(42)
    assert q.shape[0] == 1
    check_one_hot(c, pp, q[0], [4, 10 + 6], P.next_power_two(19))


def test_create_request_d3():
    # client_test.cpp:129-167: 82 items, d=3 -> dims [5, 5, 4]; index 42 -> (2, 0, 2)
    c, pp = make_client(82, 3)
    assert pp.dimensions == [5, 5, 4]
    q = c.create_query_for(42)
    assert q.shape[0] == 1
    check_one_hot(c, pp, q[0], [2, 5 + 0, 5 + 5 + 2], P.next_power_two(14))


@pytest.mark.parametrize("index,row,col,which", [(12345679, 2760, 2959, 1), (12346679, 2760, 3959, 2)])
def test_create_request_multi_dim_multi_ct(index, row, col, which):
    # client_test.cpp:169-267: 20M items, d=2 -> dims [4473, 4472], three query ciphertexts
    c, pp = make_client(20000000, 2)
    rows, cols = 4473, 4472
    assert pp.dimensions == [rows, cols]
    q = c.create_query_for(index)
    assert q.shape[0] == 3
    check_one_hot(c, pp, q[0], [row], N)
    if which == 1:
        check_one_hot(c, pp, q[1], [col + rows - N], N)
        check_one_hot(c, pp, q[2], [], 1)
    else:
        check_one_hot(c, pp, q[1], [], 1)
        check_one_hot(c, pp, q[2], [col + rows - 2 * N], P.next_power_two((rows + cols) % N))


def test_create_request_invalid_index():
    # client_test.cpp:269-272
    c, pp = make_client(100)
    with pytest.raises(pir_amd.PirGpuError) as e:
        c.CreateRequest([101])
    assert e.value.code == pir_amd.StatusCode.INVALID_ARGUMENT and "invalid index 101" in e.value.message
    with pytest.raises(pir_amd.PirGpuError):
        c.create_query_for(100)


CREATE_REQUEST = [  # client_test.cpp:321-348 (dbsize, indices, m)
    (10000, [5005], N), (10000, [0], N), (10000, [1], N), (10000, [3333], N), (10000, [4095], N),
    (10000, [4096], N), (10000, [4097], N), (10000, [8191], N), (10000, [8192], 2048), (10000, [8193], 2048),
    (10000, [9007], 2048), (10000, [9999], 2048), (4096, [0], 4096), (4096, [4095], 4096),
    (16384, [12288], 4096), (16384, [12289], 4096), (16384, [16383], 4096), (10000, [0, 8191], N),
    (10000, [0, 5005, 8191], N), (10000, [0, 1, 2, 3, 4, 5], N),
]


@pytest.mark.parametrize("dbsize,indices,m", CREATE_REQUEST)
def test_create_request_table(dbsize, indices, m):
    # client_test.cpp:278-319
    c, pp = make_client(dbsize)
    queries, galois, _ = split_request(c.CreateRequest(indices))
    assert len(queries) == len(indices) and galois[0]
    for query, desired in zip(queries, indices):
        assert len(query) == dbsize // N + 1
        for ct in query:
            if desired is None or desired >= N:
                if desired is not None:
                    desired -= N
                check_one_hot(c, pp, ct, [], 1)
            else:
                check_one_hot(c, pp, ct, [desired], m)
                desired = None


# ---------------------------------------------------------------- response decode (client_test.cpp:350-515)

def decomp_ct(c, ct, d):
    """ProcessResponseTest::DecompCT (client_test.cpp:372-389)."""
    cts = [ct]
    for _ in range(d - 1):
        cts = [c.encrypt(pt) for one in cts for pt in c.reencode(one)]
    return cts


def save_response(pp, replies):
    pid = data_pid(pp)
    out = b""
    for cts in replies:
        out += seal_wire._field(1, b"".join(seal_wire._field(1, seal_wire.save_ciphertext(ct, pid, False))
                                            for ct in cts))
    return out


RESPONSES = [(1000, d, 64, 7680, [720, 777, 839], [0, 3648, 7616]) for d in (1, 2, 3)] + \
            [(1000, 4, 64, 7680, [777], [3648])]      # d=4: 1728 ciphertexts per reply, one value keeps it quick


@pytest.mark.parametrize("dbsize,d,elem,pt_size,indices,offsets", RESPONSES)
def test_process_response(dbsize, d, elem, pt_size, indices, offsets):
    # client_test.cpp:397-424 (table :505-515)
    c, pp = make_client(dbsize, d, elem)
    bits = int(np.log2(pp.encryption_parameters.plain_modulus))
    rng = np.random.default_rng(99)
    values = [rng.integers(0, 256, size=pt_size, dtype=np.uint8).tobytes() for _ in indices]
    replies = []
    for v in values:
        rc, pt = oracle.string_encode(v, bits, N)
        assert rc == 0
        replies.append(decomp_ct(c, c.encrypt(pt), d))
    assert len(replies[0]) == c.reply_ct_count
    got = c.ProcessResponse(indices, save_response(pp, replies))
    assert got == [v[o:o + elem] for v, o in zip(values, offsets)]


def test_process_response_ct_multiply_mode():
    # client_test.cpp:426-453: one ciphertext per reply when use_ciphertext_multiplication is set
    c, pp = make_client(1000, 2, 64, use_ct_mult=True)
    bits = int(np.log2(pp.encryption_parameters.plain_modulus))
    rng = np.random.default_rng(5)
    v = rng.integers(0, 256, size=7680, dtype=np.uint8).tobytes()
    rc, pt = oracle.string_encode(v, bits, N)
    ct = c.encrypt(pt)
    assert c.ProcessResponse([777], save_response(pp, [[ct]])) == [v[3648:3648 + 64]]
    with pytest.raises(pir_amd.PirGpuError) as e:
        c.ProcessResponse([777], save_response(pp, [[ct, ct]]))
    assert e.value.code == pir_amd.StatusCode.INVALID_ARGUMENT


def integer_encode(value, t):
    """seal::IntegerEncoder::encode(int64) (base 2): bits of |value|, negated coefficients for value < 0."""
    mag, coeffs = abs(value), []
    while mag:
        coeffs.append((1 if value > 0 else t - 1) if mag & 1 else 0)
        mag >>= 1
    return coeffs or [0]


@pytest.mark.parametrize("d", [1, 2, 3])
def test_process_response_integer(d):
    # client_test.cpp:455-480
    c, pp = make_client(1000, d, 64)
    t = pp.encryption_parameters.plain_modulus
    rng = np.random.default_rng(7)
    values = [int(v) for v in rng.integers(-2**63, 2**63 - 1, size=3, dtype=np.int64)] + [0, -1, 2**63 - 1, -2**63]
    replies = [decomp_ct(c, c.encrypt(integer_encode(v, t)), d) for v in values]
    assert c.ProcessResponseInteger(save_response(pp, replies)) == values


def test_process_response_errors():
    c, pp = make_client(1000, 2, 64)
    ct = c.encrypt([1])
    good = decomp_ct(c, ct, 2)
    with pytest.raises(pir_amd.PirGpuError) as e:        # client.cpp:163-166
        c.ProcessResponse([1, 2], save_response(pp, [good]))
    assert e.value.code == pir_amd.StatusCode.INVALID_ARGUMENT and "Number of indexes" in e.value.message
    with pytest.raises(pir_amd.PirGpuError) as e:        # client.cpp:229-232
        c.ProcessResponse([1], save_response(pp, [good[:-1]]))
    assert e.value.code == pir_amd.StatusCode.INVALID_ARGUMENT and "does not match expected" in e.value.message
    with pytest.raises(pir_amd.PirGpuError) as e:        # corrupt SEAL object -> LoadCiphertexts fails
        blob = bytearray(save_response(pp, [good]))
        blob[20] ^= 0xFF
        c.ProcessResponse([1], bytes(blob))
    assert e.value.code == pir_amd.StatusCode.INVALID_ARGUMENT
    with pytest.raises(pir_amd.PirGpuError):
        c.string_decode(np.zeros(N, dtype=np.uint64), 64, N * 15 // 8 - 63)


def test_string_decode_matches_oracle_model():
    c, pp = make_client(1000, 1, 64)
    bits = int(np.log2(pp.encryption_parameters.plain_modulus))
    rng = np.random.default_rng(3)
    data = rng.integers(0, 256, size=7680, dtype=np.uint8).tobytes()
    rc, pt = oracle.string_encode(data, bits, N)
    for off, ln in [(0, 64), (1, 7), (3648, 64), (7616, 64), (13, 1), (0, 7680)]:
        assert c.string_decode(pt, ln, off) == data[off:off + ln]


# ---------------------------------------------------------------- against the CPU oracle server

@pytest.mark.parametrize("dbsize,d,elem,indexes", [(87, 1, 64, [0, 42, 86]), (300, 2, 128, [7, 299]),
                                                   (600, 3, 256, [123])])
def test_round_trip_through_oracle_server(dbsize, d, elem, indexes):
    """correctness_test.cpp:95-113 with the product client and the oracle as the server: the request's
    ciphertexts and Galois keys (parsed from the wire bytes) must drive an independent implementation of
    expansion / multiply to replies the client decodes to the database items."""
    enc = P.generate_encryption_params(N, 20)
    pp = P.create_pir_parameters(dbsize, elem, d, enc)
    c = pir_amd.PIRClient.Create(pp, seed=b"oracle-rt")
    op = oracle.create_pir_parameters(dbsize, elem, d, N=N, plain_bits=20)
    assert to_product_params(op).dimensions == pp.dimensions
    orc = oracle.Oracle.from_params(op)
    raw = generate_test_db(dbsize, elem)
    rc, db_ntt = orc.db_encode(raw.tobytes(), dbsize, elem, op.items_per_plaintext, op.eff_bits_per_coeff, op.num_pt)
    assert rc == 0

    request = c.CreateRequest(indexes)
    queries, galois, _ = split_request(request)
    # Galois keys from the wire == the residue-level accessor
    keys = {}
    pid = seal_wire.parms_id(N, enc.coeff_modulus, enc.plain_modulus)
    body = galois[0]
    assert body[16:48] == pid
    for g in P.generate_galois_elts(N):
        keys[g] = c.galois_key(g)
    # default: seed-compressed Serializable<GaloisKeys> like the reference client (client.cpp:47-54) -- the
    # independent Python loader (hashlib BLAKE2b inside BLAKE2Xb) must re-expand it to exactly these keys
    assert len(galois[0]) < 0.55 * len(seal_wire.save_galois_keys(keys, N, pid))
    loaded = seal_wire.load_kswitch_keys(galois[0], enc.coeff_modulus, N)
    assert sorted(2 * i + 1 for i in loaded) == sorted(keys)
    for i, key in loaded.items():
        assert np.array_equal(key, keys[2 * i + 1])
    # fully expanded objects on request: byte-identical to the Python model's encoding
    c.set_seeded_keys(False)
    _, galois_x, _ = split_request(c.CreateRequest(indexes))
    assert galois_x[0] == seal_wire.save_galois_keys(keys, N, pid)
    c.set_seeded_keys(True)

    replies = []
    for q in queries:
        rc, reply = orc.process_query(db_ntt, op.dimensions, np.stack(q), keys)
        assert rc == 0
        replies.append(reply)
    items = c.ProcessResponse(indexes, save_response(pp, replies))
    assert items == [raw[i].tobytes() for i in indexes]


def test_request_passes_server_side_validation():
    c, pp = make_client(5000, 2, 32)
    req = c.CreateRequest([1, 4999])
    lib = capi.load()
    import ctypes as C
    p = capi.make_params(pp)
    n = C.c_uint32()
    buf = (C.c_uint8 * len(req)).from_buffer_copy(req)
    lib.pirgpu_wire_validate_request.argtypes = [C.POINTER(capi.Params), C.POINTER(C.c_uint8), C.c_size_t,
                                                 C.POINTER(C.c_uint32)]
    lib.pirgpu_wire_validate_request.restype = C.c_int
    assert lib.pirgpu_wire_validate_request(C.byref(p), buf, len(req), C.byref(n)) == 0
    assert n.value == 2


# ---------------------------------------------------------------- the oracle's own client, same tables

ORACLE_LAYOUT = [  # (dbsize, d, index, [(ct, slot, m)])  client_test.cpp:67-267,321-348
    (100, 1, 5, [(0, 5, 128)]),
    (82, 2, 42, [(0, 4, 32), (0, 16, 32)]),
    (82, 3, 42, [(0, 2, 16), (0, 5, 16), (0, 12, 16)]),
    (20000000, 2, 12345679, [(0, 2760, N), (1, 2959 + 4473 - N, N)]),
    (20000000, 2, 12346679, [(0, 2760, N), (2, 3959 + 4473 - 2 * N, 1024)]),
    (10000, 1, 8192, [(2, 0, 2048)]),
    (10000, 1, 4097, [(1, 1, N)]),
    (16384, 1, 16383, [(3, 4095, 4096)]),
]


@pytest.mark.parametrize("dbsize,d,index,expected", ORACLE_LAYOUT)
def test_oracle_client_query_layout(dbsize, d, index, expected):
    """The test-side client (oracle/client.py) follows the same reference tables as the product client."""
    from oracle.client import Client
    op = oracle.create_pir_parameters(dbsize, 0, d, N=N, plain_bits=16)
    orc = oracle.Oracle.from_params(op)
    cl = Client(orc, seed=3)
    q = cl.create_query_for(op, index)
    assert q.shape[0] == sum(op.dimensions) // N + 1
    want = {}
    for ct, slot, m in expected:
        want.setdefault(ct, {})[slot] = m
    for c in range(q.shape[0]):
        pt = cl.decrypt(q[c])
        assert set(np.nonzero(pt)[0].tolist()) == set(want.get(c, {}))
        for slot, m in want.get(c, {}).items():
            assert int(pt[slot]) * m % op.t == 1
