#!/usr/bin/env python3
"""Generates tests/golden/cfg1_n2048.npz: one complete (parameters, raw database, Galois keys,
query, reply) tuple for BASELINE.json configs[0] (N=2048, 1 data prime + special prime,
DB = 2^10 x 32 B, d=1), produced by the CPU oracle with fixed seeds.

The reference itself cannot be run here (SEAL is absent) and holds no ciphertext fixtures, so this
vector pins the ORACLE (and through it the GPU path) against regressions; the reference's
plaintext-level known answers live in tests/test_oracle_known_answers.py as transcribed tables.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from pir_fixtures import PirSetup  # noqa: E402


def main():
    moduli = oracle.coeff_modulus_create(2048, [27, 27])
    t = oracle.plain_modulus_batching(2048, 14)
    s = PirSetup(1 << 10, 32, 1, N=2048, moduli=moduli, t=t, seed=42, client_seed=2026)
    index = 777
    query = s.client.create_query_for(s.params, index)
    rc, reply = s.orc.process_query(s.db_ntt, s.params.dimensions, query, s.galois_keys)
    assert rc == 0
    elts = sorted(s.galois_keys)
    np.savez_compressed(
        os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfg1_n2048.npz"),
        N=np.uint64(2048), moduli=np.array(moduli, dtype=np.uint64), t=np.uint64(t),
        num_items=np.uint64(1 << 10), bytes_per_item=np.uint64(32), dimensions=np.array(s.params.dimensions),
        raw=s.raw, galois_elts=np.array(elts, dtype=np.uint32),
        galois_keys=np.stack([s.galois_keys[g] for g in elts]), index=np.uint64(index), query=query, reply=reply,
        db_ntt_first=s.db_ntt[0], db_ntt_last=s.db_ntt[-1])
    print("wrote cfg1_n2048.npz; reply sha:", __import__("hashlib").sha256(reply.tobytes()).hexdigest()[:16])


if __name__ == "__main__":
    main()
