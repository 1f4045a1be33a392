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


# ---- seed-compressed objects (SEAL 3.5.6 Serializable<>): BLAKE2Xb PRNG + uniform sampler ----
# Independent of pir_amd/csrc/wire_codec.cpp: the BLAKE2b core here is hashlib's (the BLAKE2 team's reference
# code inside CPython); BLAKE2X is expressed through its parameter block (xof_length = upper half of hashlib's
# 64-bit node_offset).

SEED_BYTES = 64


_B2_IV = np.array([0x6A09E667F3BCC908, 0xBB67AE8584CAA73B, 0x3C6EF372FE94F82B, 0xA54FF53A5F1D36F1,
                   0x510E527FADE682D1, 0x9B05688C2B3E6C1F, 0x1F83D9ABFB41BD6B, 0x5BE0CD19137E2179], dtype=np.uint64)
_B2_SIGMA = [[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15], [14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3],
             [11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4], [7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8],
             [9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13], [2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9],
             [12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11], [13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10],
             [6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5], [10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0]]


def _rotr(x, n):
    return (x >> np.uint64(n)) | (x << np.uint64(64 - n))


def _b2_compress(h, m, t, last):
    """RFC 7693 compression F on `lanes` independent states at once: h [8, lanes], m [16, lanes] (uint64)."""
    v = np.concatenate([h, np.repeat(_B2_IV[:, None], h.shape[1], axis=1)])
    v[12] ^= np.uint64(t)
    if last:
        v[14] = ~v[14]

    def g(a, b, c, d, x, y):
        v[a] = v[a] + v[b] + x
        v[d] = _rotr(v[d] ^ v[a], 32)
        v[c] = v[c] + v[d]
        v[b] = _rotr(v[b] ^ v[c], 24)
        v[a] = v[a] + v[b] + y
        v[d] = _rotr(v[d] ^ v[a], 16)
        v[c] = v[c] + v[d]
        v[b] = _rotr(v[b] ^ v[c], 63)

    with np.errstate(over="ignore"):
        for r in range(12):
            sg = _B2_SIGMA[r % 10]
            g(0, 4, 8, 12, m[sg[0]], m[sg[1]])
            g(1, 5, 9, 13, m[sg[2]], m[sg[3]])
            g(2, 6, 10, 14, m[sg[4]], m[sg[5]])
            g(3, 7, 11, 15, m[sg[6]], m[sg[7]])
            g(0, 5, 10, 15, m[sg[8]], m[sg[9]])
            g(1, 6, 11, 12, m[sg[10]], m[sg[11]])
            g(2, 7, 8, 13, m[sg[12]], m[sg[13]])
            g(3, 4, 9, 14, m[sg[14]], m[sg[15]])
    return h ^ v[:8] ^ v[8:]


def b2_param(digest_length, key_length=0, fanout=1, depth=1, leaf_length=0, node_offset=0, xof_length=0, node_depth=0,
             inner_length=0, salt=b"", person=b"") -> np.ndarray:
    """The 64-byte BLAKE2b parameter block (BLAKE2X splits the 8-byte node offset into node_offset | xof_length)."""
    blk = struct.pack("<BBBBIIIBB14s16s16s", digest_length, key_length, fanout, depth, leaf_length, node_offset,
                      xof_length, node_depth, inner_length, b"", salt, person)
    return np.frombuffer(blk, dtype="<u8").astype(np.uint64)


def blake2b_param(params: np.ndarray, data: bytes, key: bytes = b"") -> bytes:
    """BLAKE2b of `data` under each of `lanes` parameter blocks (params [lanes, 8]); returns lanes x 64 bytes."""
    lanes = params.shape[0]
    h = (_B2_IV[:, None] ^ params.T).astype(np.uint64)
    msg = (key.ljust(128, b"\0") if key else b"") + data
    blocks = [msg[i:i + 128] for i in range(0, len(msg), 128)] or [b""]
    t = 0
    for i, b in enumerate(blocks):
        last = i == len(blocks) - 1
        t += len(b)
        m = np.frombuffer(b.ljust(128, b"\0"), dtype="<u8").astype(np.uint64)
        h = _b2_compress(h, np.repeat(m[:, None], lanes, axis=1), t, last)
    return np.ascontiguousarray(h.T).astype("<u8").tobytes()


def blake2xb(outlen: int, data: bytes, key: bytes = b"") -> bytes:
    """BLAKE2Xb (blake2xb.c of the BLAKE2 reference code, vendored by SEAL 3.5.6): root = BLAKE2b-512 with
    xof_length set; output block i = BLAKE2b(root) with fanout = depth = 0, leaf = inner = 64, node_offset = i."""
    h0 = blake2b_param(b2_param(64, len(key), 1, 1, xof_length=outlen)[None, :], data, key)[:64]
    nblk = (outlen + 63) // 64
    params = np.stack([b2_param(min(64, outlen - 64 * i), 0, 0, 0, leaf_length=64, node_offset=i, xof_length=outlen,
                                inner_length=64) for i in range(nblk)])
    full = blake2b_param(params, h0)
    return b"".join(full[64 * i: 64 * i + min(64, outlen - 64 * i)] for i in range(nblk))


class SealPrng:
    """BlakePRNG: 4096-byte buffers blake2xb(4096, counter_le64, key=seed), counter = 0, 1, ..."""

    def __init__(self, seed: bytes):
        assert len(seed) == SEED_BYTES
        self.seed, self.counter, self.buf, self.head = seed, 0, b"", 0

    def u32(self) -> int:
        if self.head == len(self.buf):
            self.buf = blake2xb(4096, struct.pack("<Q", self.counter), self.seed)
            self.counter += 1
            self.head = 0
        (v,) = struct.unpack_from("<I", self.buf, self.head)
        self.head += 4
        return v


def sample_poly_uniform(seed: bytes, moduli, N: int) -> np.ndarray:
    rng = SealPrng(seed)
    max_random = 0x7FFFFFFFFFFFFFFF
    out = np.empty((len(moduli), N), dtype=np.uint64)
    for j, q in enumerate(moduli):
        max_multiple = max_random - max_random % q - 1
        for i in range(N):
            while True:
                a = rng.u32()
                b = rng.u32()
                r = (a << 31) | (b >> 1)
                if r < max_multiple:
                    break
            out[j, i] = r % q
    return out


def save_ciphertext_seeded(c0: np.ndarray, seed: bytes, pid: bytes, is_ntt: bool) -> bytes:
    """c0: [nres, N]; the c1 half is replaced by the seed it is re-sampled from."""
    nres, N = c0.shape
    body = pid + struct.pack("<BQQQd", 1 if is_ntt else 0, 2, N, nres, 1.0) + save_intarray(c0.reshape(-1)) + seed
    return header(16 + len(body)) + body


def save_galois_keys_seeded(keys: dict, seeds: dict, N: int, key_pid: bytes) -> bytes:
    """keys: {elt: ndarray[k, 2, k+1, N]} whose [:, 1] halves equal sample_poly_uniform(seeds[elt][j], ...)."""
    dim1 = max((g - 1) // 2 for g in keys) + 1 if keys else 0
    body = key_pid + struct.pack("<Q", dim1)
    by_index = {(g - 1) // 2: g for g in keys}
    for index in range(dim1):
        g = by_index.get(index)
        if g is None:
            body += struct.pack("<Q", 0)
            continue
        key = keys[g]
        body += struct.pack("<Q", key.shape[0])
        for j in range(key.shape[0]):
            ct = save_ciphertext_seeded(key[j, 0], seeds[g][j], key_pid, True)
            body += header(16 + len(ct)) + ct
    return header(16 + len(body)) + body


def load_kswitch_keys(buf: bytes, moduli, N: int):
    """KSwitchKeys (expanded or seeded entries) -> {index: ndarray[k, 2, k+1, N]}"""
    magic, hs, _, _, compr, _, total = struct.unpack_from("<HBBBBHQ", buf, 0)
    assert magic == SEAL_MAGIC and total == len(buf) and compr == 0
    off = 16 + 32
    (dim1,) = struct.unpack_from("<Q", buf, off)
    off += 8
    km = len(moduli)
    out = {}
    for index in range(dim1):
        (dim2,) = struct.unpack_from("<Q", buf, off)
        off += 8
        if not dim2:
            continue
        key = np.empty((dim2, 2, km, N), dtype=np.uint64)
        for j in range(dim2):
            _, _, _, _, _, _, pk_total = struct.unpack_from("<HBBBBHQ", buf, off)
            ct = buf[off + 16: off + pk_total]
            (count,) = struct.unpack_from("<Q", ct, 48 + 33 + 16)
            data = np.frombuffer(ct, dtype="<u8", count=count, offset=48 + 33 + 24)
            if count == km * N:      # seeded
                seed = ct[48 + 33 + 24 + 8 * count: 48 + 33 + 24 + 8 * count + SEED_BYTES]
                key[j, 0] = data.reshape(km, N)
                key[j, 1] = sample_poly_uniform(seed, moduli, N)
            else:
                key[j] = data.reshape(2, km, N)
            off += pk_total
        out[index] = key
    return out


# ---- EncryptionParameters (SEAL 3.5.6 save_members) and pir.PIRParameters (payload.proto:45-69) ----

def save_modulus(v: int) -> bytes:
    return header(24) + struct.pack("<Q", v)


def save_encryption_parameters(N: int, moduli, t: int) -> bytes:
    """scheme (u8, BFV = 1), poly_modulus_degree, coeff_modulus_size, each Modulus (own header + u64), plain_modulus."""
    body = struct.pack("<BQQ", 1, N, len(moduli)) + b"".join(save_modulus(q) for q in moduli) + save_modulus(t)
    return header(16 + len(body)) + body


def load_encryption_parameters(buf: bytes):
    magic, hs, _, _, compr, _, total = struct.unpack_from("<HBBBBHQ", buf, 0)
    assert magic == SEAL_MAGIC and hs == 16 and compr == 0 and total == len(buf), "not an uncompressed SEAL object"
    scheme, N, n = struct.unpack_from("<BQQ", buf, 16)
    assert scheme == 1, "BFV expected"
    off = 16 + 17
    vals = []
    for _ in range(n + 1):
        m, _, _, _, _, _, tot = struct.unpack_from("<HBBBBHQ", buf, off)
        assert m == SEAL_MAGIC and tot == 24
        vals.append(struct.unpack_from("<Q", buf, off + 16)[0])
        off += 24
    return N, vals[:-1], vals[-1]


def _varint_field(num, v):
    return _varint(num << 3) + _varint(v)


def save_pir_parameters(num_items, num_pt, dimensions, enc_bytes, bytes_per_item, items_per_plaintext,
                        bits_per_coeff=0, use_ct_mult=False) -> bytes:
    out = _varint_field(1, num_items) if num_items else b""
    out += _field(2, b"".join(_varint(d) for d in dimensions))          # packed repeated uint32
    out += _field(3, enc_bytes)
    out += _varint_field(4, num_pt) if num_pt else b""
    for num, v in ((5, bytes_per_item), (6, items_per_plaintext), (7, bits_per_coeff), (8, 1 if use_ct_mult else 0)):
        if v:
            out += _varint_field(num, v)
    return out


def _read_varint(buf, i):
    v = shift = 0
    while True:
        b = buf[i]
        i += 1
        v |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            return v, i


def load_pir_parameters(buf: bytes) -> dict:
    out = {"num_items": 0, "num_pt": 0, "dimensions": [], "encryption_parameters": b"", "bytes_per_item": 0,
           "items_per_plaintext": 0, "bits_per_coeff": 0, "use_ciphertext_multiplication": False}
    names = {1: "num_items", 4: "num_pt", 5: "bytes_per_item", 6: "items_per_plaintext", 7: "bits_per_coeff"}
    i = 0
    while i < len(buf):
        tag, i = _read_varint(buf, i)
        num, wt = tag >> 3, tag & 7
        if wt == 0:
            v, i = _read_varint(buf, i)
            if num == 2:
                out["dimensions"].append(v)          # unpacked encoding
            elif num == 8:
                out["use_ciphertext_multiplication"] = bool(v)
            elif num in names:
                out[names[num]] = v
        elif wt == 2:
            ln, i = _read_varint(buf, i)
            payload = buf[i:i + ln]
            i += ln
            if num == 2:
                j = 0
                while j < len(payload):
                    v, j = _read_varint(payload, j)
                    out["dimensions"].append(v)
            elif num == 3:
                out["encryption_parameters"] = bytes(payload)
        else:
            raise ValueError("unexpected wire type %d in PIRParameters" % wt)
    return out


def load_request(buf: bytes, moduli, N: int):
    """serialized pir.Request -> (queries: list of ndarray [nq, 2, k, N], galois keys {elt: key}, relin bytes)"""
    queries, keys, relin = [], {}, b""
    for num, payload in _parse(buf):
        if num == 1:
            cts = [load_ciphertext(bytes(p))[2] for n2, p in _parse(payload) if n2 == 1]
            queries.append(np.stack(cts))
        elif num == 2:
            keys = {2 * i + 1: k for i, k in load_kswitch_keys(bytes(payload), moduli, N).items()}
        elif num == 3:
            relin = bytes(payload)
    return queries, keys, relin


def save_response(replies, data_pid: bytes) -> bytes:
    """replies: list of ndarray [n, 2, k, N] -> serialized pir.Response"""
    out = b""
    for r in replies:
        out += _field(1, b"".join(_field(1, save_ciphertext(ct, data_pid, False)) for ct in r))
    return out
