#!/usr/bin/env python3
"""Replays a (PIRParameters, database, Request, Response) tuple produced by the REFERENCE (OpenMined/PIR on a
machine that has Microsoft SEAL 3.5.6) and reports whether this repository reproduces the Response byte for byte
(the one check that can close "bit-exact vs SEAL", which no test inside this image can).

    python tools/check_external_pair.py params.bin db.bin request.bin response.bin [--no-gpu]

  params.bin    pir::PIRParameters::SerializeAsString()                 (payload.proto:45-69)
  db.bin        the raw database items, num_items x bytes_per_item bytes, item-major
                (what PIRDatabase::Create(vector<string>, params) was given, database.cpp:52-58)
  request.bin   pir::Request::SerializeAsString()  as PIRClient::CreateRequest made it (client.cpp:80-90)
  response.bin  pir::Response::SerializeAsString() as PIRServer::ProcessRequest returned it (server.cpp:44-65)

Two legs, each compared with response.bin:
  oracle  tests/seal_wire.py (Python codec) -> oracle/ (CPU restatement) -> Python codec   [always]
  gpu     the raw request bytes through pirgpu_process_request (C++ codec + HIP kernels)    [when a GPU is present]
On a mismatch the first differing reply / ciphertext / polynomial / residue / coefficient is printed.
INTEGRATION.md section 5 shows the dozen lines to add to the reference's benchmark or tests to dump the four files.
Exit status: 0 = every leg that ran reproduced the response, 1 = mismatch, 2 = input could not be parsed.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def first_difference(got: bytes, want: bytes, W):
    if got == want:
        return None
    try:
        g, w = W.load_response(got), W.load_response(want)
    except Exception as e:       # not even the framing agrees
        n = next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), min(len(got), len(want)))
        return "serialized responses differ at byte %d (lengths %d / %d); re-parse failed: %r" % (n, len(got), len(want), e)
    if len(g) != len(w):
        return "reply count %d != %d" % (len(g), len(w))
    for r, (a, b) in enumerate(zip(g, w)):
        if a.shape != b.shape:
            return "reply %d: shape %s != %s" % (r, a.shape, b.shape)
        if not np.array_equal(a, b):
            idx = tuple(int(x) for x in np.argwhere(a != b)[0])
            return ("reply %d, ciphertext %d, polynomial %d, residue %d, coefficient %d: got %d, reference %d "
                    "(%d of %d words differ)" % ((r,) + idx + (int(a[idx]), int(b[idx]), int((a != b).sum()), a.size)))
    return "residues agree but the serialized bytes differ (framing / header fields)"


def run(params_b: bytes, db_b: bytes, request_b: bytes, response_b: bytes, use_gpu: bool = True, log=print) -> int:
    import oracle
    import seal_wire as W
    try:
        pp = W.load_pir_parameters(params_b)
        N, moduli, t = W.load_encryption_parameters(pp["encryption_parameters"])
    except Exception as e:
        log("cannot parse PIRParameters / EncryptionParameters: %r" % (e,))
        return 2
    log("N=%d, moduli=%s (last = key-switching special prime), t=%d, items=%d x %d B, num_pt=%d, dimensions=%s"
        % (N, [hex(q) for q in moduli], t, pp["num_items"], pp["bytes_per_item"], pp["num_pt"], pp["dimensions"]))
    if pp["use_ciphertext_multiplication"]:
        log("use_ciphertext_multiplication = true is outside the scope of this repository")
        return 2
    if len(db_b) != pp["num_items"] * pp["bytes_per_item"]:
        log("db.bin has %d bytes, expected num_items * bytes_per_item = %d" % (len(db_b), pp["num_items"] * pp["bytes_per_item"]))
        return 2
    failed = False
    # ---- oracle leg
    p = oracle.PirParams(N=N, moduli=moduli, t=t, num_items=pp["num_items"], num_pt=pp["num_pt"],
                         dimensions=pp["dimensions"], bytes_per_item=pp["bytes_per_item"],
                         items_per_plaintext=pp["items_per_plaintext"], bits_per_coeff=pp["bits_per_coeff"])
    orc = oracle.Oracle.from_params(p)
    try:
        queries, keys, _ = W.load_request(request_b, moduli, N)
    except Exception as e:
        log("cannot parse Request: %r" % (e,))
        return 2
    rc, db = orc.db_encode(db_b, p.num_items, p.bytes_per_item, p.items_per_plaintext, p.eff_bits_per_coeff, p.num_pt)
    if rc:
        log("oracle db_encode failed with status %d" % rc)
        return 2
    replies = []
    for q in queries:
        rc, rep = orc.process_query(db, p.dimensions, q, keys)
        if rc:
            log("oracle process_query failed with status %d" % rc)
            return 1
        replies.append(rep)
    got = W.save_response(replies, W.parms_id(N, moduli[:-1], t))
    diff = first_difference(got, response_b, W)
    log("oracle leg: %s" % ("response reproduced byte for byte (%d bytes)" % len(got) if diff is None else "MISMATCH: " + diff))
    failed |= diff is not None
    # ---- GPU leg
    if use_gpu:
        try:
            import pir_amd
            from pir_amd.parameters import EncryptionParams, PIRParameters
            ppar = PIRParameters(num_items=p.num_items, num_pt=p.num_pt, dimensions=list(p.dimensions),
                                 encryption_parameters=EncryptionParams(N, list(moduli), t),
                                 bytes_per_item=p.bytes_per_item, items_per_plaintext=p.items_per_plaintext,
                                 bits_per_coeff=p.bits_per_coeff)
            raw = np.frombuffer(db_b, dtype=np.uint8).reshape(p.num_items, p.bytes_per_item)
            dbh = pir_amd.PIRDatabase.Create(ppar, raw)
            srv = pir_amd.PIRServer.Create(dbh, ppar)
            got = srv.ProcessRequest(request_b)
            diff = first_difference(got, response_b, W)
            log("gpu leg: %s" % ("response reproduced byte for byte (%d bytes)" % len(got) if diff is None else "MISMATCH: " + diff))
            failed |= diff is not None
        except Exception as e:
            log("gpu leg skipped / failed: %r" % (e,))
            if "no HIP device" not in repr(e):
                failed = True
    return 1 if failed else 0


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("params")
    ap.add_argument("db")
    ap.add_argument("request")
    ap.add_argument("response")
    ap.add_argument("--no-gpu", action="store_true")
    a = ap.parse_args()
    blobs = [open(f, "rb").read() for f in (a.params, a.db, a.request, a.response)]
    sys.exit(run(*blobs, use_gpu=not a.no_gpu))


if __name__ == "__main__":
    main()
