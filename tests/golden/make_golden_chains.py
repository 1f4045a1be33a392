#!/usr/bin/env python3
"""Writes tests/golden/chains.json with the CPU oracle (see chains.py).  Run from the repo root:
python tests/golden/make_golden_chains.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import chains  # noqa: E402
import oracle  # noqa: E402


def main():
    out = {}
    for name in chains.CASES:
        p, raw, keys, query = chains.make_inputs(name, oracle)
        orc = oracle.Oracle.from_params(p)
        rc, db = orc.db_encode(raw.tobytes(), p.num_items, p.bytes_per_item, p.items_per_plaintext,
                               p.eff_bits_per_coeff, p.num_pt)
        assert rc == 0
        rc, reply = orc.process_query(db, p.dimensions, query, keys)
        assert rc == 0
        out[name] = chains.digest(p, raw, keys, query, reply)
        print(name, p.dimensions, reply.shape, out[name]["reply_sha256"][:16])
    json.dump(out, open(chains.JSON, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
