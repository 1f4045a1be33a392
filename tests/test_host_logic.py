"""CPU-side checks of the product's host logic: the C-ABI library loads and exports every
symbol include/pirgpu.h declares, fails loudly without a GPU, the parameter mirror matches
the reference's tables, and the wire codec's hashing matches an independent implementation."""
import ctypes as C
import hashlib
import os
import re

import numpy as np
import pytest

import oracle
import pir_amd
from pir_amd import capi, parameters as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "pirgpu.h")).read()
    declared = set(re.findall(r"\b(pirgpu_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"pirgpu_ctx", "pirgpu_params"}
    assert len(declared) >= 30
    lib = capi.load()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(capi.SIGNATURES), declared ^ set(capi.SIGNATURES)


def test_struct_layout_matches_header():
    # sizeof(pirgpu_params) as the C compiler lays it out (gcc, same ABI as hipcc's host side)
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "s.c")
        open(src, "w").write('#include <stdio.h>\n#include "pirgpu.h"\nint main(){printf("%zu %zu %zu", '
                             'sizeof(pirgpu_params), __builtin_offsetof(pirgpu_params, num_pt), '
                             '__builtin_offsetof(pirgpu_params, shard_begin));return 0;}')
        exe = os.path.join(d, "s")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        size, off_pt, off_shard = map(int, subprocess.run([exe], capture_output=True, text=True).stdout.split())
    assert C.sizeof(capi.Params) == size
    assert capi.Params.num_pt.offset == off_pt and capi.Params.shard_begin.offset == off_shard


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    pp = P.create_pir_parameters(10, 0, 1, P.generate_encryption_params(4096, 20))
    with pytest.raises(pir_amd.PirGpuError) as e:
        pir_amd.PIRDatabase.Create(pp)
    assert e.value.code == pir_amd.StatusCode.INTERNAL and "no HIP device" in e.value.message


def test_parameter_mirror_matches_reference_tables():
    # parameters_test.cpp:47-98
    p = P.create_pir_parameters(1026, 256)
    assert (p.num_pt, p.items_per_plaintext, p.dimensions) == (27, 38, [27])
    p = P.create_pir_parameters(19011, 500, 3)
    assert (p.num_pt, p.items_per_plaintext, p.dimensions) == (1001, 19, [11, 10, 10])
    p = P.create_pir_parameters(77412, 777, 2, P.generate_encryption_params(8192), True, 12)
    assert (p.num_pt, p.items_per_plaintext, p.dimensions, p.bits_per_coeff) == (5161, 15, [72, 72], 12)
    with pytest.raises(ValueError):
        P.create_pir_parameters(10, 0, 1, P.generate_encryption_params(4096, 20), False, 25)
    with pytest.raises(ValueError):
        P.create_pir_parameters(10, 20000, 1)


@pytest.mark.parametrize("dbsize,elem,d,N,bits", [(1 << 16, 288, 1, 4096, 24), (1 << 20, 288, 2, 4096, 24),
                                                  (500, 0, 2, 4096, 24), (87, 0, 2, 4096, 16), (82, 0, 3, 4096, 16)])
def test_parameter_mirror_matches_oracle(dbsize, elem, d, N, bits):
    a = P.create_pir_parameters(dbsize, elem, d, P.generate_encryption_params(N, bits))
    b = oracle.create_pir_parameters(dbsize, elem, d, N=N, plain_bits=bits)
    assert (a.num_pt, a.items_per_plaintext, a.bytes_per_item, a.dimensions) == \
           (b.num_pt, b.items_per_plaintext, b.bytes_per_item, b.dimensions)
    assert a.encryption_parameters.coeff_modulus == b.moduli and a.encryption_parameters.plain_modulus == b.t
    for idx in (0, dbsize // 3, dbsize - 1):
        assert a.calculate_indices(idx) == oracle.calculate_indices(idx, b.items_per_plaintext, b.dimensions)
        assert a.calculate_item_offset(idx) == oracle.calculate_item_offset(idx, b.items_per_plaintext,
                                                                            b.bytes_per_item)


def test_index_tables():
    # database_test.cpp:409-464
    for n, size, d, index, exp in [(87, 0, 2, 42, [4, 6]), (82, 0, 3, 75, [3, 3, 3]), (5000, 64, 1, 2222, [18])]:
        p = P.create_pir_parameters(n, size, d, P.generate_encryption_params(4096, 16))
        assert p.calculate_indices(index) == exp
    for n, d, exp in [(82, 2, [10, 9]), (975, 2, [32, 31]), (1001, 3, [11, 10, 10]), (1000001, 3, [101, 100, 100])]:
        assert P.calculate_dimensions(n, d) == exp
    assert P.generate_galois_elts(4096) == oracle.generate_galois_elts(4096)
    assert [P.next_power_two(x) for x in (0, 1, 3, 324, 4096)] == [1, 1, 4, 512, 4096]


def test_wire_blake2b_and_parms_id_match_hashlib():
    lib = capi.load()
    lib.pirgpu_wire_blake2b.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    lib.pirgpu_wire_blake2b.restype = None
    rng = np.random.default_rng(0)
    for n in (0, 1, 8, 64, 127, 128, 129, 300, 4096):
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        for outlen in (32, 64):
            out = C.create_string_buffer(outlen)
            lib.pirgpu_wire_blake2b(out, outlen, data, n)
            assert out.raw == hashlib.blake2b(data, digest_size=outlen).digest(), (n, outlen)
    from seal_wire import parms_id
    lib.pirgpu_wire_parms_id.argtypes = [C.c_uint32, C.POINTER(C.c_uint64), C.c_size_t, C.c_uint64,
                                         C.POINTER(C.c_uint64)]
    lib.pirgpu_wire_parms_id.restype = None
    moduli = oracle.BFV_DEFAULT[4096]
    for mods in (moduli, moduli[:2]):
        out = (C.c_uint64 * 4)()
        lib.pirgpu_wire_parms_id(4096, (C.c_uint64 * len(mods))(*mods), len(mods), 0xFFC001, out)
        assert bytes(out) == parms_id(4096, mods, 0xFFC001)


def _wire_fixture():
    """A valid serialized pir.Request (N=2048 golden vector: small) and its pirgpu_params."""
    import seal_wire as W
    z = np.load(os.path.join(ROOT, "tests", "golden", "cfg1_n2048.npz"))
    N, moduli, t = int(z["N"]), [int(x) for x in z["moduli"]], int(z["t"])
    keys = {int(g): z["galois_keys"][i] for i, g in enumerate(z["galois_elts"])}
    gk = W.save_galois_keys(keys, N, W.parms_id(N, moduli, t))
    req = W.save_request([z["query"]], gk, W.parms_id(N, moduli[:-1], t))
    p = capi.Params()
    p.poly_modulus_degree, p.num_data_primes = N, len(moduli) - 1
    p.coeff_modulus[0], p.special_prime, p.plain_modulus = moduli[0], moduli[1], t
    p.num_dimensions, p.num_pt = 1, 10
    p.dimensions[0] = 10
    return req, p


def _validate(lib, p, data):
    n = C.c_uint32(0)
    buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data if data else b"\0")
    return lib.pirgpu_wire_validate_request(C.byref(p), buf, len(data), C.byref(n)), n.value


def test_wire_request_validation_and_fuzz():
    """The server parses untrusted client bytes: a valid Request validates, and truncations, bit flips and
    spliced garbage are rejected with a status code (never a crash, never accepted when structurally broken)."""
    lib = capi.load()
    lib.pirgpu_wire_validate_request.argtypes = [C.POINTER(capi.Params), C.POINTER(C.c_uint8), C.c_size_t,
                                                 C.POINTER(C.c_uint32)]
    lib.pirgpu_wire_validate_request.restype = C.c_int
    req, p = _wire_fixture()
    assert _validate(lib, p, req) == (0, 1)
    assert _validate(lib, p, b"")[0] == 3                       # no galois keys -> InvalidArgument
    rng = np.random.default_rng(11)
    allowed = {0, 3, 12}
    for cut in [1, 2, 7, 16, 17, 48, 100, 1000, len(req) // 2, len(req) - 9, len(req) - 1]:
        rc, _ = _validate(lib, p, req[:cut])
        assert rc in (3, 12), cut                               # truncated objects never validate
    data = bytearray(req)
    for _ in range(300):
        m = bytearray(data)
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(0, len(m)))
            m[pos] ^= 1 << int(rng.integers(0, 8))
        rc, _ = _validate(lib, p, bytes(m))
        assert rc in allowed
    for _ in range(50):                                         # random garbage of assorted lengths
        g = rng.integers(0, 256, int(rng.integers(1, 4096)), dtype=np.uint8).tobytes()
        assert _validate(lib, p, g)[0] in allowed
    # wrong parameters (different plain modulus) -> parms_id mismatch
    p2 = capi.Params.from_buffer_copy(p)
    p2.plain_modulus = 40961
    assert _validate(lib, p2, req)[0] == 3


# ---------------------------------------------------------------- seed-compressed SEAL objects

def _wire_hooks():
    lib = capi.load()
    lib.pirgpu_wire_blake2xb.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    lib.pirgpu_wire_blake2xb.restype = None
    lib.pirgpu_wire_sample_poly_uniform.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.c_uint32, C.c_uint32,
                                                    C.POINTER(C.c_uint64)]
    lib.pirgpu_wire_sample_poly_uniform.restype = None
    lib.pirgpu_wire_load_kswitch_key.argtypes = [C.POINTER(capi.Params), C.POINTER(C.c_uint8), C.c_size_t, C.c_uint64,
                                                 C.POINTER(C.c_uint64)]
    lib.pirgpu_wire_load_kswitch_key.restype = C.c_int
    lib.pirgpu_wire_validate_request.argtypes = [C.POINTER(capi.Params), C.POINTER(C.c_uint8), C.c_size_t,
                                                 C.POINTER(C.c_uint32)]
    lib.pirgpu_wire_validate_request.restype = C.c_int
    return lib


def test_blake2xb_matches_python_model():
    """C++ BLAKE2Xb (wire_codec.cpp) == the numpy model in tests/seal_wire.py, whose BLAKE2b core and parameter
    block handling are checked against hashlib field by field (hashlib itself refuses depth = 0, which BLAKE2X's
    output blocks use, so it cannot serve as the model directly)."""
    import seal_wire as W
    lib = _wire_hooks()
    rng = np.random.default_rng(5)
    for n in (0, 1, 64, 127, 128, 129, 300):               # the BLAKE2b core against hashlib, every parameter field
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        for key in (b"", b"k" * 64, b"abc"):
            for (ds, fo, de, ls, no, nd, isz) in [(64, 1, 1, 0, 0, 0, 0), (48, 2, 3, 77, 0x123456789, 1, 32),
                                                  (17, 0, 255, 64, 5, 0, 64)]:
                want = hashlib.blake2b(d, digest_size=ds, key=key, fanout=fo, depth=de, leaf_size=ls, node_offset=no,
                                       node_depth=nd, inner_size=isz).digest()
                par = W.b2_param(ds, len(key), fo, de, ls, no & 0xFFFFFFFF, no >> 32, nd, isz)
                assert W.blake2b_param(par[None, :], d, key)[:ds] == want
    for outlen in (1, 63, 64, 65, 128, 1000, 4096):
        for inlen in (0, 8, 200):
            for keylen in (0, 32, 64):
                data = rng.integers(0, 256, inlen, dtype=np.uint8).tobytes()
                key = rng.integers(0, 256, keylen, dtype=np.uint8).tobytes()
                out = C.create_string_buffer(outlen)
                lib.pirgpu_wire_blake2xb(out, outlen, data, inlen, key, keylen)
                assert out.raw == W.blake2xb(outlen, data, key), (outlen, inlen, keylen)


def test_seeded_sampler_matches_python_model():
    import seal_wire as W
    lib = _wire_hooks()
    moduli = oracle.coeff_modulus_create(2048, [27, 27]) + [oracle.BFV_DEFAULT[4096][2]]
    for seed in (bytes(64), bytes(range(64))):
        out = np.empty((3, 2048), dtype=np.uint64)
        lib.pirgpu_wire_sample_poly_uniform(seed, (C.c_uint64 * 3)(*moduli), 3, 2048, out.ctypes.data_as(capi.u64p))
        want = W.sample_poly_uniform(seed, moduli, 2048)
        assert np.array_equal(out, want)
        assert all(int(out[j].max()) < moduli[j] for j in range(3))


def test_seeded_galois_keys_load_like_expanded_ones():
    """A seed-compressed GaloisKeys object written by the PYTHON codec loads through the C++ codec to the same
    residues as its expanded twin; a seeded request validates; malformed RelinKeys are InvalidArgument
    (reference server.cpp:53-58) while well-formed ones (seeded or expanded) are accepted."""
    import seal_wire as W
    lib = _wire_hooks()
    z = np.load(os.path.join(ROOT, "tests", "golden", "cfg1_n2048.npz"))
    N, moduli, t = int(z["N"]), [int(x) for x in z["moduli"]], int(z["t"])
    _, p = _wire_fixture()
    key_pid, data_pid = W.parms_id(N, moduli, t), W.parms_id(N, moduli[:-1], t)
    rng = np.random.default_rng(9)
    keys, seeds = {}, {}
    for g in (3, N + 1, N // 2 + 1):
        key = np.array(z["galois_keys"][0])                     # [k=1, 2, 2, N]: keep c0, replace c1 by a seeded half
        seeds[g] = [rng.integers(0, 256, 64, dtype=np.uint8).tobytes()]
        key[0, 1] = W.sample_poly_uniform(seeds[g][0], moduli, N)
        keys[g] = key
    seeded = W.save_galois_keys_seeded(keys, seeds, N, key_pid)
    expanded = W.save_galois_keys(keys, N, key_pid)
    assert len(seeded) < 0.6 * len(expanded)
    for blob in (seeded, expanded):
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        for g, key in keys.items():
            out = np.empty_like(key)
            assert lib.pirgpu_wire_load_kswitch_key(C.byref(p), buf, len(blob), (g - 1) // 2,
                                                    out.ctypes.data_as(capi.u64p)) == 0
            assert np.array_equal(out, key), g
        out = np.empty_like(keys[3])
        assert lib.pirgpu_wire_load_kswitch_key(C.byref(p), buf, len(blob), 2, out.ctypes.data_as(capi.u64p)) == 5
    relin_ok = W.save_galois_keys_seeded({1: keys[3]}, {1: seeds[3]}, N, key_pid)     # RelinKeys: one entry, index 0
    req = W.save_request([z["query"]], seeded, data_pid, relin_keys=relin_ok)
    assert _validate(lib, p, req) == (0, 1)
    assert _validate(lib, p, W.save_request([z["query"]], seeded, data_pid, relin_keys=relin_ok[:-5]))[0] == 3
    assert _validate(lib, p, W.save_request([z["query"]], seeded, data_pid, relin_keys=b"\x01\x02\x03"))[0] == 3
    bad = bytearray(seeded)
    bad[-70] ^= 0x80          # inside the last c0: coefficient out of range or seed/size damage -> rejected or changed
    assert _validate(lib, p, W.save_request([z["query"]], seeded[:-1], data_pid))[0] == 3
