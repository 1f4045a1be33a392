"""Shape math / string encoder / utils of the oracle against the reference's test tables."""
import numpy as np
import pytest

import oracle


def test_create_pir_parameters_sanity():
    # parameters_test.cpp:47-60 (defaults: N=4096, 20-bit t)
    p = oracle.create_pir_parameters(1026, 256)
    assert (p.num_items, p.num_pt, p.bytes_per_item, p.items_per_plaintext, p.dimensions) == (1026, 27, 256, 38, [27])


def test_create_pir_parameters_multidim():
    # parameters_test.cpp:62-77
    p = oracle.create_pir_parameters(19011, 500, 3)
    assert (p.num_pt, p.items_per_plaintext, p.dimensions) == (1001, 19, [11, 10, 10])


def test_create_pir_parameters_all():
    # parameters_test.cpp:79-98 (N=8192, default 20-bit t, bits_per_coeff=12)
    p = oracle.create_pir_parameters(77412, 777, 2, N=8192, use_ciphertext_multiplication=True, bits_per_coeff_=12)
    assert (p.num_pt, p.items_per_plaintext, p.dimensions, p.bits_per_coeff) == (5161, 15, [72, 72], 12)


def test_baseline_config_shapes():
    # SURVEY.md section 8 table, config 2 and 3 (benchmark.cpp:17-23 parameters)
    p2 = oracle.create_pir_parameters(1 << 16, 288, 1, N=4096, plain_bits=24)
    assert (p2.items_per_plaintext, p2.num_pt, p2.dimensions) == (40, 1639, [1639])
    p3 = oracle.create_pir_parameters(1 << 20, 288, 2, N=4096, plain_bits=24)
    assert (p3.items_per_plaintext, p3.num_pt, p3.dimensions) == (40, 26215, [162, 162])


# database_test.cpp:409-427 (num_items, item size, d, index, expected) with GenerateEncryptionParams(4096, 16)
INDICES = [(100, 0, 1, 42, [42]), (100, 0, 1, 7, [7]), (84, 0, 2, 7, [0, 7]), (87, 0, 2, 27, [3, 0]),
           (87, 0, 2, 42, [4, 6]), (87, 0, 2, 86, [9, 5]), (82, 0, 3, 3, [0, 0, 3]), (82, 0, 3, 20, [1, 0, 0]),
           (82, 0, 3, 75, [3, 3, 3]), (5000, 64, 1, 2222, [18]), (5000, 64, 1, 1200, [10])]


@pytest.mark.parametrize("n,size,d,index,expected", INDICES)
def test_calculate_indices(n, size, d, index, expected):
    p = oracle.create_pir_parameters(n, size, d, N=4096, plain_bits=16)
    assert oracle.calculate_indices(index, p.items_per_plaintext, p.dimensions) == expected


# database_test.cpp:445-449
@pytest.mark.parametrize("n,size,index,expected", [(100, 0, 42, 0), (1000, 64, 42, 2688), (1000, 64, 960, 0),
                                                    (1000, 64, 999, 2496)])
def test_calculate_offset(n, size, index, expected):
    p = oracle.create_pir_parameters(n, size, 1, N=4096, plain_bits=16)
    assert oracle.calculate_item_offset(index, p.items_per_plaintext, p.bytes_per_item) == expected


# database_test.cpp:456-464
@pytest.mark.parametrize("n,d,expected", [(100, 1, [100]), (100, 2, [10, 10]), (82, 2, [10, 9]), (975, 2, [32, 31]),
                                          (1000, 3, [10, 10, 10]), (1001, 3, [11, 10, 10]),
                                          (1000001, 3, [101, 100, 100])])
def test_calculate_dimensions(n, d, expected):
    assert oracle.calculate_dimensions(n, d) == expected


def test_items_per_plaintext():
    # string_encoder_test.cpp:64-71 (N=4096, t=20 bits -> 19 bits per coefficient)
    lib = oracle.load()
    b = oracle.bits_per_coeff(oracle.plain_modulus_batching(4096, 20))
    assert b == 19
    for size, exp in [(1, 9728), (9728, 1), (9729, 0), (99999, 0), (64, 152), (288, 33)]:
        assert lib.orc_items_per_plaintext(4096, b, size) == exp


# string_encoder_test.cpp:207-211
@pytest.mark.parametrize("N,bits,exp", [(4096, 20, 9728), (4096, 16, 7680), (8192, 20, 19456)])
def test_max_bytes_per_plaintext(N, bits, exp):
    b = oracle.bits_per_coeff(oracle.plain_modulus_batching(N, bits))
    assert oracle.load().orc_max_bytes_per_plaintext(N, b) == exp


def test_string_encode_decode():
    # string_encoder_test.cpp:73-83
    value = b"This is a string test for random VALUES@!#"
    rc, coeffs = oracle.string_encode(value, 19, 4096)
    assert rc == 0 and coeffs.shape[0] == -(-len(value) * 8 // 19)
    rc, out = oracle.string_decode(coeffs, 19, len(value))
    assert rc == 0 and out == value


def test_string_encode_decode_random_and_vector():
    # string_encoder_test.cpp:85-119
    rng = np.random.default_rng(42)
    v = rng.integers(0, 256, 9728, dtype=np.uint8).tobytes()
    rc, coeffs = oracle.string_encode(v, 19, 4096)
    assert rc == 0 and coeffs.shape[0] == 4096 and int(coeffs.max()) < (1 << 19)
    assert oracle.string_decode(coeffs, 19, 9728)[1] == v
    items = [rng.integers(1, 256, 64, dtype=np.uint8).tobytes() for _ in range(152)]
    rc, coeffs = oracle.string_encode(b"".join(items), 19, 4096)
    assert rc == 0
    for i, it in enumerate(items):
        assert oracle.string_decode(coeffs, 19, 64, 64 * i) == (0, it)


def test_string_encode_too_big_and_decode_too_big():
    # string_encoder_test.cpp:121-161
    rng = np.random.default_rng(42)
    assert oracle.string_encode(rng.integers(0, 256, 9729, dtype=np.uint8).tobytes(), 19, 4096)[0] == 3
    assert oracle.string_encode(rng.integers(0, 256, 141 * 69, dtype=np.uint8).tobytes(), 19, 4096)[0] == 3
    rc, coeffs = oracle.string_encode(rng.integers(0, 256, 9728, dtype=np.uint8).tobytes(), 19, 4096)
    assert oracle.string_decode(coeffs, 19, 100, 9629)[0] == 3


def test_string_encode_python_bit_model():
    """Independent model: concatenated bits MSB-first, cut into b-bit fields, zero padded on the right."""
    rng = np.random.default_rng(7)
    for nbytes, b in [(1, 19), (5, 23), (100, 13), (288 * 3, 23), (17, 8), (9, 4)]:
        data = rng.integers(0, 256, nbytes, dtype=np.uint8).tobytes()
        bits = "".join(format(x, "08b") for x in data)
        nc = -(-len(bits) // b)
        bits = bits.ljust(nc * b, "0")
        exp = [int(bits[i * b:(i + 1) * b], 2) for i in range(nc)]
        rc, coeffs = oracle.string_encode(data, b, 4096)
        assert rc == 0 and coeffs.tolist() == exp


def test_utils_tables():
    # utils_test.cpp
    for n, e in [(0, 1), (1, 1), (2, 2), (3, 4), (8, 8), (9, 16), (4095, 4096), (4097, 8192)]:
        assert oracle.next_power_two(n) == e
    for v, e in [(1, 0), (2, 1), (3, 2), (4, 2), (5, 3), (4096, 12), (4097, 13)]:
        assert oracle.ceil_log2(v) == e
    for v, e in [(1, 0), (2, 1), (3, 1), (4, 2), (4095, 11), (4096, 12)]:
        assert oracle.log2(v) == e
    assert oracle.generate_galois_elts(4096) == [4097, 2049, 1025, 513, 257, 129, 65, 33, 17, 9, 5, 3]


def test_parameter_tables():
    # SURVEY App. A.1
    assert oracle.plain_modulus_batching(4096, 20) == 0xFC001
    assert oracle.plain_modulus_batching(4096, 16) == 40961
    assert oracle.plain_modulus_batching(4096, 24) == 0xFFC001
    assert oracle.plain_modulus_batching(16384, 24) == 0xFD0001
    assert oracle.plain_modulus_batching(2048, 14) == 12289
    for N, ms in oracle.BFV_DEFAULT.items():
        for q in ms:
            assert oracle.is_prime(q) and (q - 1) % (2 * N) == 0
