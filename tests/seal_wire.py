"""Test-side Python model of the wire format: pir/proto/payload.proto framing around SEAL
3.5.6 binary objects.  Independent of pir_amd/csrc/wire.cpp so the two
can be checked against each other."""
import hashlib
import struct

import numpy as np

SEAL_MAGIC = 0xA15E


def parms_id(N, moduli, t):
    data = struct.pack("<%dQ" % (3 + len(moduli)), 1, N, *moduli, t)
    return hashlib.blake2b(data, digest_size=32).digest()


def header(total):
    return struct.pack("<HBBBBHQ", SEAL_MAGIC, 0x10, 3, 5, 0, 0, total)


def save_intarray(words: np.ndarray) -> bytes:
    body = struct.pack("<Q", words.size) + np.ascontiguousarray(words, dtype="<u8").tobytes()
    return header(16 + len(body)) + body


def save_ciphertext(ct: np.ndarray, pid: bytes, is_ntt: bool) -> bytes:
    """ct: [2, nres, N]"""
    _, nres, N = ct.shape
    body = pid + struct.pack("<BQQQd", 1 if is_ntt else 0, 2, N, nres, 1.0) + save_intarray(ct.reshape(-1))
    return header(16 + len(body)) + body


def load_ciphertext(buf: bytes):
    magic, hs, vmaj, vmin, compr, _, total = struct.unpack_from("<HBBBBHQ", buf, 0)
    assert magic == SEAL_MAGIC and hs == 16 and compr == 0 and total == len(buf)
    pid = buf[16:48]
    is_ntt, size, N, nres, scale = struct.unpack_from("<BQQQd", buf, 48)
    off = 48 + 33
    amagic, _, _, _, _, _, atotal = struct.unpack_from("<HBBBBHQ", buf, off)
    assert amagic == SEAL_MAGIC and off + atotal == len(buf)
    (count,) = struct.unpack_from("<Q", buf, off + 16)
    data = np.frombuffer(buf, dtype="<u8", count=count, offset=off + 24).reshape(size, nres, N).copy()
    return pid, bool(is_ntt), data


def save_galois_keys(keys: dict, N: int, key_pid: bytes) -> bytes:
    """keys: {galois_elt: ndarray[k, 2, k+1, N]} -> KSwitchKeys layout."""
    dim1 = max((g - 1) // 2 for g in keys) + 1 if keys else 0
    body = key_pid + struct.pack("<Q", dim1)
    by_index = {(g - 1) // 2: v for g, v in keys.items()}
    for index in range(dim1):
        key = by_index.get(index)
        if key is None:
            body += struct.pack("<Q", 0)
            continue
        body += struct.pack("<Q", key.shape[0])
        for j in range(key.shape[0]):
            ct = save_ciphertext(key[j], key_pid, True)
            body += header(16 + len(ct)) + ct          # PublicKey wrapper
    return header(16 + len(body)) + body


# ---- proto3 (payload.proto) ----

def _varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _field(num, payload: bytes):
    return _varint((num << 3) | 2) + _varint(len(payload)) + payload


def _parse(buf):
    i, out = 0, []
    while i < len(buf):
        tag = 0
        shift = 0
        while True:
            b = buf[i]
            i += 1
            tag |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                break
        assert tag & 7 == 2
        ln = 0
        shift = 0
        while True:
            b = buf[i]
            i += 1
            ln |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                break
        out.append((tag >> 3, buf[i:i + ln]))
        i += ln
    return out


def save_request(queries, galois_keys_bytes: bytes, data_pid: bytes, relin_keys: bytes = b"") -> bytes:
    """queries: list of ndarray [nq, 2, k, N] -> serialized pir.Request (payload.proto:27-36)."""
    out = b""
    for q in queries:
        cts = b"".join(_field(1, save_ciphertext(ct, data_pid, False)) for ct in q)
        out += _field(1, cts)
    out += _field(2, galois_keys_bytes)
    if relin_keys:
        out += _field(3, relin_keys)
    return out


def load_response(buf: bytes):
    """serialized pir.Response (payload.proto:39-42) -> list of ndarray [n, 2, k, N]"""
    replies = []
    for num, payload in _parse(buf):
        assert num == 1
        cts = [load_ciphertext(bytes(p))[2] for n2, p in _parse(payload) if n2 == 1]
        replies.append(np.stack(cts))
    return replies
