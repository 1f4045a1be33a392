"""tools/check_external_pair.py on a self-made tuple: the product client's Request (seed-compressed keys, like a
reference client's), the oracle's replies serialized by the Python codec as the "reference" Response.  A SEAL
machine supplies the real thing; this keeps the tool itself from rotting."""
import os
import sys

import numpy as np
import pytest

import oracle
import pir_amd
from pir_amd import parameters as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_external_pair as tool  # noqa: E402
import seal_wire as W  # noqa: E402


def _tuple(dbsize=900, d=2, elem=64, indexes=(3, 888)):
    N = 4096
    enc = P.generate_encryption_params(N, 20)
    pp = P.create_pir_parameters(dbsize, elem, d, enc)
    c = pir_amd.PIRClient.Create(pp, seed=b"external-pair")
    raw = np.random.default_rng(5).integers(0, 256, size=(dbsize, elem), dtype=np.uint8)
    request = c.CreateRequest(list(indexes))
    params_b = W.save_pir_parameters(pp.num_items, pp.num_pt, pp.dimensions,
                                     W.save_encryption_parameters(N, enc.coeff_modulus, enc.plain_modulus),
                                     pp.bytes_per_item, pp.items_per_plaintext, pp.bits_per_coeff)
    op = oracle.create_pir_parameters(dbsize, elem, d, N=N, plain_bits=20)
    orc = oracle.Oracle.from_params(op)
    rc, db = orc.db_encode(raw.tobytes(), dbsize, elem, op.items_per_plaintext, op.eff_bits_per_coeff, op.num_pt)
    assert rc == 0
    queries, keys, _ = W.load_request(request, enc.coeff_modulus, N)
    replies = [orc.process_query(db, op.dimensions, q, keys)[1] for q in queries]
    response = W.save_response(replies, W.parms_id(N, enc.coeff_modulus[:-1], enc.plain_modulus))
    assert c.ProcessResponse(list(indexes), response) == [raw[i].tobytes() for i in indexes]
    return params_b, raw.tobytes(), request, response


def test_params_codec_round_trip():
    enc = W.save_encryption_parameters(4096, oracle.BFV_DEFAULT[4096], 0xFFC001)
    assert W.load_encryption_parameters(enc) == (4096, oracle.BFV_DEFAULT[4096], 0xFFC001)
    b = W.save_pir_parameters(1 << 20, 26215, [162, 162], enc, 288, 40, 0)
    got = W.load_pir_parameters(b)
    assert (got["num_items"], got["num_pt"], got["dimensions"], got["bytes_per_item"], got["items_per_plaintext"]) == \
        (1 << 20, 26215, [162, 162], 288, 40) and got["encryption_parameters"] == enc


def test_oracle_leg_accepts_and_localises_a_flipped_bit():
    params_b, db_b, request, response = _tuple()
    lines = []
    assert tool.run(params_b, db_b, request, response, use_gpu=False, log=lines.append) == 0
    assert any("reproduced byte for byte" in ln for ln in lines)
    bad = bytearray(response)
    bad[len(bad) // 2 + 5000] ^= 1          # inside the coefficient data of a reply ciphertext
    lines = []
    assert tool.run(params_b, db_b, request, bytes(bad), use_gpu=False, log=lines.append) == 1
    assert any("MISMATCH" in ln and "coefficient" in ln for ln in lines), lines
    assert tool.run(params_b[:40], db_b, request, response, use_gpu=False, log=lines.append) == 2


@pytest.mark.gpu
def test_gpu_leg_reproduces_the_response():
    params_b, db_b, request, response = _tuple()
    lines = []
    assert tool.run(params_b, db_b, request, response, use_gpu=True, log=lines.append) == 0, lines
    assert any(ln.startswith("gpu leg: response reproduced") for ln in lines), lines
