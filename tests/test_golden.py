"""Golden vector tests/golden/cfg1_n2048.npz (BASELINE.json configs[0], made by tests/golden/make_golden.py):
the oracle must keep reproducing it (CPU) and the GPU path must reproduce it through the C ABI (GPU)."""
import os

import numpy as np
import pytest

import oracle

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg1_n2048.npz")


def _load():
    z = np.load(G)
    N, moduli, t = int(z["N"]), [int(x) for x in z["moduli"]], int(z["t"])
    params = oracle.create_pir_parameters(int(z["num_items"]), int(z["bytes_per_item"]), 1, N=N, moduli=moduli, t=t)
    keys = {int(g): z["galois_keys"][i] for i, g in enumerate(z["galois_elts"])}
    return z, params, keys


def test_oracle_reproduces_golden_reply():
    z, p, keys = _load()
    assert p.dimensions == list(z["dimensions"]) == [10] and p.items_per_plaintext == 104
    o = oracle.Oracle.from_params(p)
    rc, db = o.db_encode(z["raw"].tobytes(), p.num_items, p.bytes_per_item, p.items_per_plaintext,
                         p.eff_bits_per_coeff, p.num_pt)
    assert rc == 0
    assert np.array_equal(db[0], z["db_ntt_first"]) and np.array_equal(db[-1], z["db_ntt_last"])
    rc, reply = o.process_query(db, p.dimensions, z["query"], keys)
    assert rc == 0 and np.array_equal(reply, z["reply"])


@pytest.mark.gpu
def test_gpu_reproduces_golden_reply():
    import pir_amd
    from gpu_helpers import to_product_params
    z, p, keys = _load()
    pp = to_product_params(p)
    db = pir_amd.PIRDatabase.Create(pp, z["raw"])
    srv = pir_amd.PIRServer.Create(db, pp)
    assert np.array_equal(db.read_plaintext(0), z["db_ntt_first"])
    assert np.array_equal(db.read_plaintext(p.num_pt - 1), z["db_ntt_last"])
    srv.set_galois_keys(keys)
    assert np.array_equal(srv.process_query(z["query"]), z["reply"])
    db.close()
