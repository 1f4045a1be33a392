"""ctypes binding of libpirgpu.so (include/pirgpu.h).  No torch types cross this boundary."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PIRGPU_LIB", os.path.join(HERE, "libpirgpu.so"))   # override: A/B builds (tools/build_variant.py)

MAX_PRIMES, MAX_DIMS = 8, 8
OK, INVALID_ARGUMENT, FAILED_PRECONDITION, UNIMPLEMENTED, INTERNAL = 0, 3, 9, 12, 13

u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)


class Params(C.Structure):
    _fields_ = [
        ("poly_modulus_degree", C.c_uint32),
        ("num_data_primes", C.c_uint32),
        ("coeff_modulus", C.c_uint64 * MAX_PRIMES),
        ("special_prime", C.c_uint64),
        ("plain_modulus", C.c_uint64),
        ("num_dimensions", C.c_uint32),
        ("dimensions", C.c_uint32 * MAX_DIMS),
        ("num_pt", C.c_uint64),
        ("num_items", C.c_uint64),
        ("bytes_per_item", C.c_uint32),
        ("items_per_plaintext", C.c_uint32),
        ("bits_per_coeff", C.c_uint32),
        ("use_ciphertext_multiplication", C.c_uint32),
        ("device", C.c_int32),
        ("shard_begin", C.c_uint32),
        ("shard_end", C.c_uint32),
        ("slot_begin", C.c_uint32),
        ("slot_end", C.c_uint32),
    ]


def make_params(params, device: int = 0, shard=None, slots=None) -> Params:
    """PIRParameters (+ device / first-dimension shard / slot shard) -> the pirgpu_params struct of include/pirgpu.h."""
    enc = params.encryption_parameters
    p = Params()
    p.poly_modulus_degree = enc.poly_modulus_degree
    p.num_data_primes = len(enc.coeff_modulus) - 1
    for i, q in enumerate(enc.coeff_modulus[:-1]):
        p.coeff_modulus[i] = q
    p.special_prime = enc.coeff_modulus[-1] if len(enc.coeff_modulus) > 1 else 0
    p.plain_modulus = enc.plain_modulus
    p.num_dimensions = len(params.dimensions)
    for i, d in enumerate(params.dimensions):
        p.dimensions[i] = d
    p.num_pt = params.num_pt
    p.num_items = params.num_items
    p.bytes_per_item = params.bytes_per_item
    p.items_per_plaintext = params.items_per_plaintext
    p.bits_per_coeff = params.bits_per_coeff
    p.use_ciphertext_multiplication = 1 if params.use_ciphertext_multiplication else 0
    p.device = device
    if shard is not None:
        b, e = int(shard[0]), int(shard[1])
        if b == e:                      # empty shard: (0, 0) means "whole database" in the C ABI
            b = e = params.dimensions[0]
        p.shard_begin, p.shard_end = b, e
    if slots is not None:
        p.slot_begin, p.slot_end = int(slots[0]), int(slots[1])
    return p


# every symbol include/pirgpu.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "pirgpu_create": (C.c_int, [C.POINTER(Params), C.POINTER(C.c_void_p)]),
    "pirgpu_destroy": (None, [C.c_void_p]),
    "pirgpu_last_error": (C.c_char_p, [C.c_void_p]),
    "pirgpu_create_error": (C.c_char_p, []),
    "pirgpu_db_load_items": (C.c_int, [C.c_void_p, u8p, C.c_uint64, C.c_uint32]),
    "pirgpu_db_load_coeffs": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, u64p]),
    "pirgpu_db_size": (C.c_uint64, [C.c_void_p]),
    "pirgpu_db_read_plaintext": (C.c_int, [C.c_void_p, C.c_uint64, u64p]),
    "pirgpu_db_finalize": (C.c_int, [C.c_void_p, C.c_int]),
    "pirgpu_set_transparent_policy": (C.c_int, [C.c_void_p, C.c_int]),
    "pirgpu_zero_plaintexts": (C.c_uint64, [C.c_void_p]),
    "pirgpu_set_remote_zero_plaintexts": (C.c_int, [C.c_void_p, C.c_uint64]),
    "pirgpu_check_ready": (C.c_int, [C.c_void_p]),
    "pirgpu_ntt_mode": (C.c_int, [C.c_void_p]),
    "pirgpu_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "pirgpu_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]),
    "pirgpu_packed_selector_bytes": (C.c_uint64, [C.c_void_p]),
    "pirgpu_batch_expand_packed": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                             C.POINTER(C.c_uint32), C.c_uint32]),
    "pirgpu_batch_run_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "pirgpu_slots_packed_bytes": (C.c_uint64, [C.c_void_p, C.c_uint32]),
    "pirgpu_slots_expand_async": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                            C.POINTER(C.c_uint32), C.c_uint32, C.c_void_p, C.c_void_p]),
    "pirgpu_slots_scan_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "pirgpu_slots_finish_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32),
                                            C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pirgpu_set_galois_key": (C.c_int, [C.c_void_p, C.c_uint32, u64p]),
    "pirgpu_clear_galois_keys": (C.c_int, [C.c_void_p]),
    "pirgpu_set_keyset_capacity": (C.c_int, [C.c_void_p, C.c_uint32]),
    "pirgpu_keyset_lookup": (C.c_int, [C.c_void_p, u8p, C.c_size_t, C.c_int, C.POINTER(C.c_uint32)]),
    "pirgpu_keyset_verify": (C.c_int, [C.c_void_p, C.c_uint32, u8p, C.c_size_t]),
    "pirgpu_keyset_claim": (C.c_int, [C.c_void_p, u8p, C.c_size_t, C.POINTER(C.c_uint32)]),
    "pirgpu_keyset_release": (C.c_int, [C.c_void_p, C.c_uint32]),
    "pirgpu_keyset_set_key": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, u64p]),
    "pirgpu_keyset_set_keys": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_void_p)]),
    "pirgpu_query_use_keyset": (C.c_int, [C.c_void_p, C.c_uint32]),
    "pirgpu_batch_set_keysets": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]),
    "pirgpu_batch_stage_async": (C.c_int, [C.c_void_p, u64p, C.c_uint32, C.c_uint32]),
    "pirgpu_batch_unstage": (C.c_int, [C.c_void_p]),
    "pirgpu_batch_select": (C.c_int, [C.c_void_p, C.c_uint32]),
    "pirgpu_keyset_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "pirgpu_process_requests": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "pirgpu_request_error": (C.c_char_p, [C.c_uint32]),
    "pirgpu_process_requests_begin": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                                C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_int),
                                                C.POINTER(C.c_void_p)]),
    "pirgpu_process_requests_end": (C.c_int, [C.c_void_p]),
    "pirgpu_host_query_buffer": (C.c_void_p, [C.c_void_p, C.c_uint32]),
    "pirgpu_host_reply_buffer": (C.c_void_p, [C.c_void_p, C.c_uint32]),
    "pirgpu_process_query": (C.c_int, [C.c_void_p, u64p, C.c_uint32, u64p, C.c_uint64, u64p]),
    "pirgpu_reply_ct_count": (C.c_uint64, [C.c_void_p]),
    "pirgpu_expansion_ratio": (C.c_uint32, [C.c_void_p]),
    "pirgpu_query_stage": (C.c_int, [C.c_void_p, u64p, C.c_uint32]),
    "pirgpu_query_stage_async": (C.c_int, [C.c_void_p, u64p, C.c_uint32]),
    "pirgpu_query_run": (C.c_int, [C.c_void_p]),
    "pirgpu_query_fetch": (C.c_int, [C.c_void_p, u64p, C.c_uint64, u64p]),
    "pirgpu_sync": (C.c_int, [C.c_void_p]),
    "pirgpu_device_synchronize": (C.c_int, [C.c_void_p]),
    "pirgpu_set_concurrency": (C.c_int, [C.c_void_p, C.c_uint32]),
    "pirgpu_batch_stage": (C.c_int, [C.c_void_p, u64p, C.c_uint32, C.c_uint32]),
    "pirgpu_batch_run": (C.c_int, [C.c_void_p]),
    "pirgpu_batch_fetch": (C.c_int, [C.c_void_p, u64p, C.c_uint64, u64p]),
    "pirgpu_batch_reply_copy_to_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "pirgpu_batch_expand": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "pirgpu_stream_handle": (C.c_void_p, [C.c_void_p]),
    "pirgpu_join": (C.c_int, [C.c_void_p]),
    "pirgpu_join_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pirgpu_pack40_supported": (C.c_int, [C.c_void_p]),
    "pirgpu_pack40_device_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "pirgpu_unpack40_device_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "pirgpu_batch_set_host_replies": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "pirgpu_batch_next_host_replies": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "pirgpu_batch_set_reply_buffer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "pirgpu_fork": (C.c_int, [C.c_void_p]),
    "pirgpu_batch_expand_packed_async": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                                   C.POINTER(C.c_uint32), C.c_uint32]),
    "pirgpu_batch_reply_copy_to_device_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "pirgpu_reduce_fixup_device_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "pirgpu_batch_run_selectors": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "pirgpu_expand": (C.c_int, [C.c_void_p, u64p, C.c_uint32, u64p]),
    "pirgpu_expand_multi": (C.c_int, [C.c_void_p, u64p, C.c_uint32, C.c_uint64, u64p]),
    "pirgpu_substitute_power_x": (C.c_int, [C.c_void_p, u64p, C.c_uint32]),
    "pirgpu_multiply_inverse_power_of_x": (C.c_int, [C.c_void_p, u64p, C.c_uint32, u64p]),
    "pirgpu_multiply": (C.c_int, [C.c_void_p, u64p, C.c_uint64, u64p, C.c_uint64, u64p]),
    "pirgpu_ntt_forward": (C.c_int, [C.c_void_p, u64p, C.c_uint64, C.c_int]),
    "pirgpu_ntt_inverse": (C.c_int, [C.c_void_p, u64p, C.c_uint64, C.c_int]),
    "pirgpu_reduce_fixup_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "pirgpu_reply_device_ptr": (C.c_void_p, [C.c_void_p]),
    "pirgpu_process_request": (C.c_int, [C.c_void_p, u8p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "pirgpu_free": (None, [C.c_void_p]),
    "pirgpu_last_timings": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
    "pirgpu_set_error": (None, [C.c_void_p, C.c_char_p]),
    "pirgpu_get_params": (C.c_int, [C.c_void_p, C.POINTER(Params)]),
    "pirgpu_reply_copy_to_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "pirgpu_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "pirgpu_batch_scan_timings": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "pirgpu_scan_bytes": (C.c_uint64, [C.c_void_p]),
    "pirgpu_scan_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
}

_lib = None


def load() -> C.CDLL:
    """Load libpirgpu.so; fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "pir_amd: %s is missing -- build the HIP extension first (python -c 'import __graft_entry__ as g; "
            "g.build()' or python pir_amd/build.py). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


# ---------------------------------------------------------------------------------------------------
# libpirclient.so (include/pirclient.h): CPU-only client library, no device dependency.
CLIENT_LIB_PATH = os.environ.get("PIR_CLIENT_LIB", os.path.join(HERE, "libpirclient.so"))   # override: sanitizer builds
i64p = C.POINTER(C.c_int64)

CLIENT_SIGNATURES = {
    "pirclient_create": (C.c_int, [C.POINTER(Params), u8p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "pirclient_destroy": (None, [C.c_void_p]),
    "pirclient_last_error": (C.c_char_p, [C.c_void_p]),
    "pirclient_create_error": (C.c_char_p, []),
    "pirclient_create_request": (C.c_int, [C.c_void_p, u64p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "pirclient_process_response": (C.c_int, [C.c_void_p, u64p, C.c_size_t, u8p, C.c_size_t, u8p, C.c_size_t]),
    "pirclient_process_response_integer": (C.c_int, [C.c_void_p, u8p, C.c_size_t, i64p, C.c_size_t,
                                                      C.POINTER(C.c_size_t)]),
    "pirclient_free": (None, [C.c_void_p]),
    "pirclient_set_seeded_keys": (C.c_int, [C.c_void_p, C.c_int]),
    "pirclient_save_request": (C.c_int, [C.c_void_p, u64p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "pirclient_load_response": (C.c_int, [C.c_void_p, u8p, C.c_size_t, u64p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "pirclient_query_ct_count": (C.c_uint32, [C.c_void_p]),
    "pirclient_create_query": (C.c_int, [C.c_void_p, C.c_uint64, u64p, C.c_size_t, C.POINTER(C.c_uint32)]),
    "pirclient_galois_key": (C.c_int, [C.c_void_p, C.c_uint32, u64p]),
    "pirclient_process_reply": (C.c_int, [C.c_void_p, u64p, C.c_size_t, u64p]),
    "pirclient_reply_ct_count": (C.c_uint64, [C.c_void_p]),
    "pirclient_encrypt": (C.c_int, [C.c_void_p, u64p, C.c_size_t, u64p]),
    "pirclient_decrypt": (C.c_int, [C.c_void_p, u64p, u64p]),
    "pirclient_noise_budget": (C.c_int, [C.c_void_p, u64p, C.POINTER(C.c_int)]),
    "pirclient_reencode": (C.c_int, [C.c_void_p, u64p, u64p, C.c_size_t, C.POINTER(C.c_uint32)]),
    "pirclient_string_decode": (C.c_int, [C.c_void_p, u64p, C.c_size_t, C.c_size_t, u8p]),
}

_client_lib = None


def load_client() -> C.CDLL:
    """Load libpirclient.so (g++-built, runs without a GPU)."""
    global _client_lib
    if _client_lib is not None:
        return _client_lib
    if not os.path.exists(CLIENT_LIB_PATH):
        from . import build as _build       # host-only g++ build, a few seconds
        _build.build_client()
    lib = C.CDLL(CLIENT_LIB_PATH)
    for name, (res, args) in CLIENT_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _client_lib = lib
    return lib
