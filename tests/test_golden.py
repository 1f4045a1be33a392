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


# ---- BASELINE.json configs 2-5 modulus chains at small item counts (tests/golden/chains.py) ----

import sys  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import chains  # noqa: E402


def _check_reply(name, p, raw, keys, query, reply):
    want = chains.load()[name]
    got = chains.digest(p, raw, keys, query, reply)
    for key in ("dimensions", "num_pt", "items_per_plaintext", "raw_sha256", "keys_sha256", "query_sha256"):
        assert got[key] == want[key], (name, key)       # the deterministic inputs are the committed ones
    assert got["reply_shape"] == want["reply_shape"]
    assert got["reply_head"] == want["reply_head"], name
    assert got["reply_sha256"] == want["reply_sha256"], name


@pytest.mark.parametrize("name", sorted(chains.CASES))
def test_oracle_reproduces_chain_golden(name):
    p, raw, keys, query = chains.make_inputs(name, oracle)
    o = oracle.Oracle.from_params(p)
    rc, db = o.db_encode(raw.tobytes(), p.num_items, p.bytes_per_item, p.items_per_plaintext, p.eff_bits_per_coeff,
                         p.num_pt)
    assert rc == 0
    rc, reply = o.process_query(db, p.dimensions, query, keys)
    assert rc == 0
    _check_reply(name, p, raw, keys, query, reply)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(chains.CASES))
def test_gpu_reproduces_chain_golden(name):
    """The GPU path against the committed digests alone (no oracle call): configs 2-5 modulus chains, incl. the
    6- and 7-digit int8-MFMA scans (cfg 4 / cfg 5 have >= 8 rows) and the wide-fp64 NTT flavour (cfg 5)."""
    import pir_amd
    from gpu_helpers import to_product_params
    p, raw, keys, query = chains.make_inputs(name, oracle)
    pp = to_product_params(p)
    db = pir_amd.PIRDatabase.Create(pp, raw)
    srv = pir_amd.PIRServer.Create(db, pp)
    srv.set_galois_keys(keys)
    reply = srv.process_query(query)
    _check_reply(name, p, raw, keys, query, reply)
    # the batch pipeline (grouped expansion, shared database pass) must give the same bits
    srv.set_concurrency(3)
    srv.stage_batch(np.stack([query, query, query]))
    srv.run_batch()
    for r in srv.fetch_batch():
        assert np.array_equal(r, reply)
    db.close()
