"""Deterministic synthetic inputs + golden reply digests for the BASELINE.json modulus chains (configs 2-5 at
small item counts).  TEST INFRASTRUCTURE.

Inputs come from a SplitMix64 stream written out here in integer numpy ops (no dependence on numpy's or any
library's PRNG), so the committed digests stay reproducible on any box: database bytes, uniformly random
Galois-key residues and query residues (parity is a residue-level property, the query need not decrypt).
`tests/golden/chains.json` (made by `python tests/golden/make_golden_chains.py` with the CPU oracle) holds,
per case, SHA-256 of the inputs and of the full reply plus the first 16 coefficients of every reply polynomial.
"""
import hashlib
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
JSON = os.path.join(HERE, "chains.json")

BFV_DEFAULT = {
    4096: [0xFFFFEE001, 0xFFFFC4001, 0x1FFFFE0001],
    8192: [0x7FFFFFD8001, 0x7FFFFFC8001, 0xFFFFFFFC001, 0xFFFFFF6C001, 0xFFFFFEBC001],
    16384: [0xFFFFFFFD8001, 0xFFFFFFFA0001, 0xFFFFFFF00001, 0x1FFFFFFF68001, 0x1FFFFFFF50001,
            0x1FFFFFFEE8001, 0x1FFFFFFEA0001, 0x1FFFFFFE88001, 0x1FFFFFFE48001],
}

# name -> (N, moduli (data primes + special), plain bits, items, bytes per item, dimensions d)
CASES = {
    "cfg2_n4096_d1": (4096, BFV_DEFAULT[4096], 24, 600, 288, 1),
    "cfg3_n4096_d2": (4096, BFV_DEFAULT[4096], 24, 2000, 288, 2),
    "cfg4_n8192_k3_d2": (8192, BFV_DEFAULT[8192][:3] + [BFV_DEFAULT[8192][4]], 24, 1500, 1024, 2),
    "cfg5_n16384_k4_d2": (16384, BFV_DEFAULT[16384][:4] + [BFV_DEFAULT[16384][8]], 24, 11000, 288, 2),
}


def splitmix64(seed: int, n: int) -> np.ndarray:
    """n outputs of SplitMix64 started at `seed` (vectorised: output i uses state seed + (i+1)*gamma)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def residues(seed, shape, q):
    return (splitmix64(seed, int(np.prod(shape))) % np.uint64(q)).reshape(shape)


def make_inputs(name, oracle):
    N, moduli, pbits, items, bpi, d = CASES[name]
    t = oracle.plain_modulus_batching(N, pbits)
    p = oracle.create_pir_parameters(items, bpi, d, N=N, moduli=moduli, t=t)
    k = len(moduli) - 1
    raw = (splitmix64(1, items * bpi) >> np.uint64(56)).astype(np.uint8).reshape(items, bpi)
    keys = {}
    for lvl, g in enumerate(oracle.generate_galois_elts(N)):
        key = np.empty((k, 2, k + 1, N), dtype=np.uint64)
        for i in range(k + 1):
            key[:, :, i, :] = residues(1000 + 16 * lvl + i, (k, 2, N), moduli[i])
        keys[g] = key
    nq = p.dim_sum // N + 1
    query = np.empty((nq, 2, k, N), dtype=np.uint64)
    for j in range(k):
        query[:, :, j, :] = residues(77 + j, (nq, 2, N), moduli[j])
    return p, raw, keys, query


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def digest(p, raw, keys, query, reply):
    h = hashlib.sha256()
    for g in sorted(keys):
        h.update(np.ascontiguousarray(keys[g]).tobytes())
    return {"dimensions": [int(x) for x in p.dimensions], "num_pt": int(p.num_pt),
            "items_per_plaintext": int(p.items_per_plaintext), "raw_sha256": sha(raw), "keys_sha256": h.hexdigest(),
            "query_sha256": sha(query), "reply_shape": list(reply.shape), "reply_sha256": sha(reply),
            "reply_head": reply[..., :16].reshape(-1).tolist()}


def load():
    return json.load(open(JSON))
