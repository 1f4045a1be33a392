"""BASELINE.json configs[1], [3], [4] at FULL size on the GPU (configs[2] = the headline workload is
test_gpu_parity.py::test_full_size_benchmark_workload).

What can be checked at these sizes in seconds:
  * cfg 2 (2^16 x 288 B, d = 1, 1 639 plaintexts, the 2 047-node expansion tree): the whole reply against the CPU oracle;
  * cfg 4 (2^22 x 1 KiB, N = 8192, 3 data primes, 428 x 427) and cfg 5 (2^24 x 288 B, N = 16384, 4 data primes,
    321 x 321): a full oracle pass over 36 / 54 GB is minutes of CPU, so the oracle is run on a ROW SUBSET: a context
    created with shard = (0, 16) on the full database produces the partial reply of those 16 rows at the real column
    count (the 6- / 7-digit int8-MFMA scan with its column chunks, the fused upper level), and the oracle computes the
    same summand from those rows alone (cost proportional to the rows: database.cpp:170-258 is a sum over dimension 0).
    Size-independent properties on the whole database: the batch pipeline returns the same bits as single queries
    (at cfg 4 these take different scan kernels: int8-MFMA for batches, 64-bit multiply-accumulate for a single
    query), and the partial replies of a row partition sum (mod q) to the unsharded reply.
Queries and Galois keys are uniform residues (parity is a residue-level property; nothing here needs to decrypt).
"""
import numpy as np
import pytest

import oracle
import pir_amd
from gpu_helpers import random_ct, random_key, to_product_params
from pir_fixtures import oracle_partial_reply

pytestmark = pytest.mark.gpu


def _keys(orc, rng):
    N = orc.N
    return {(N >> j) + 1: random_key(orc, rng) for j in range(N.bit_length() - 1)}


def test_cfg2_full_size_d1():
    N, n_items = 4096, 1 << 16
    params = oracle.create_pir_parameters(n_items, 288, 1, N=N, plain_bits=24)
    assert params.dimensions == [1639] and params.num_pt == 1639
    orc = oracle.Oracle.from_params(params)
    rng = np.random.default_rng(216)
    raw = rng.integers(0, 256, size=(n_items, 288), dtype=np.uint8)
    keys = _keys(orc, rng)
    queries = random_ct(orc, rng, 3)[:, None]
    pp = to_product_params(params)
    db = pir_amd.PIRDatabase.Create(pp, raw)
    srv = pir_amd.PIRServer.Create(db, pp)
    srv.set_galois_keys(keys)
    info = srv.scan_info()
    assert not info["mfma"] and info["rows"] == 1 and info["cols"] == 1639      # d = 1: the 64-bit multiply-accumulate scan
    single = [srv.process_query(q) for q in queries]
    batch = srv.process_batch(queries, n_workers=3)
    for i in range(3):
        assert np.array_equal(batch[i], single[i]), i
    rc, db_ntt = orc.db_encode(raw.tobytes(), n_items, 288, params.items_per_plaintext, params.eff_bits_per_coeff,
                               params.num_pt)
    assert rc == 0
    rc, exp = orc.process_query(db_ntt, params.dimensions, queries[0], keys)
    assert rc == 0 and np.array_equal(single[0], exp)
    # plaintext shards (d = 1 shards dimension 0 = the plaintexts): partial replies sum to the reply
    acc = np.zeros_like(single[1])
    for lo, hi in [(0, 205), (205, 1024), (1024, 1639)]:
        dbs = pir_amd.PIRDatabase.Create(pp, raw, shard=(lo, hi))
        ss = pir_amd.PIRServer(dbs, pp)
        ss.set_galois_keys(keys)
        acc += ss.process_query(queries[1])
        dbs.close()
    for j, qj in enumerate(orc.moduli[: orc.k]):
        acc[:, :, j, :] %= np.uint64(qj)
    assert np.array_equal(acc, single[1])
    db.close()


def _large_d2(N, moduli, n_items, item_bytes, dims, digits, seed):
    params = oracle.create_pir_parameters(n_items, item_bytes, 2, N=N, moduli=moduli, plain_bits=24)
    assert params.dimensions == dims
    orc = oracle.Oracle.from_params(params)
    rng = np.random.default_rng(seed)
    raw = rng.integers(0, 256, size=(n_items, item_bytes), dtype=np.uint8)
    keys = _keys(orc, rng)
    queries = random_ct(orc, rng, 2)[:, None]
    pp = to_product_params(params)
    n1 = dims[1]

    # (1) a 16-row shard at the real column count against the oracle on those rows only
    cut = 16
    dbs = pir_amd.PIRDatabase.Create(pp, raw, shard=(0, cut))
    ss = pir_amd.PIRServer(dbs, pp)
    ss.set_galois_keys(keys)
    info = ss.scan_info()
    assert info["mfma"] and info["digits"] == digits and info["rows"] == cut and info["cols"] == n1, info
    part_lo = ss.process_query(queries[0])
    part_lo_batch = ss.process_batch(np.stack([queries[0], queries[1]]), n_workers=2)
    assert np.array_equal(part_lo_batch[0], part_lo)
    dbs.close()
    ipp = params.items_per_plaintext
    shard_items = min(n_items, cut * n1 * ipp)
    sub = oracle.create_pir_parameters(shard_items, item_bytes, 1, N=N, moduli=moduli, plain_bits=24)
    rc, db_rows = orc.db_encode(raw[:shard_items].tobytes(), shard_items, item_bytes, ipp, params.eff_bits_per_coeff,
                                sub.num_pt)
    assert rc == 0 and sub.num_pt == cut * n1
    rc, sv = orc.oblivious_expansion_multi(queries[0], params.dim_sum, keys)
    assert rc == 0
    rc, exp = oracle_partial_reply(orc, db_rows, dims, 0, cut, sv)
    assert rc == 0 and np.array_equal(part_lo, exp)
    del db_rows, sv

    # (2) the rest of the partition: partial replies sum to the unsharded reply
    dbs = pir_amd.PIRDatabase.Create(pp, raw, shard=(cut, dims[0]))
    ss = pir_amd.PIRServer(dbs, pp)
    ss.set_galois_keys(keys)
    part_hi = ss.process_query(queries[0])
    dbs.close()
    acc = part_lo + part_hi
    for j, qj in enumerate(orc.moduli[: orc.k]):
        acc[:, :, j, :] %= np.uint64(qj)

    # (3) the whole database: single query == batch == sum of the shards
    db = pir_amd.PIRDatabase.Create(pp, raw)
    srv = pir_amd.PIRServer.Create(db, pp)
    srv.set_galois_keys(keys)
    info = srv.scan_info()
    assert info["mfma"] and info["digits"] == digits and info["rows"] == dims[0], info
    single = srv.process_query(queries[0])
    assert np.array_equal(single, acc)
    batch = srv.process_batch(queries, n_workers=2)
    assert np.array_equal(batch[0], single)
    assert np.array_equal(batch[1], srv.process_query(queries[1]))
    db.close()

    # (4) SLOT shards (the multi-GPU step of DESIGN.md section 7.1) at the real size: two contexts, each holding half of
    # the NTT slots of every plaintext, created one after the other (together they would not fit next to their lanes at
    # cfg 5).  Expansion and finish do not depend on which slots a context holds, so context 0 expands both "ranks'"
    # queries and context 1 finishes both; each scans ITS slots for both queries with the 4-wave scan kernel of these
    # rings; the all-to-alls are tensor copies.  Replies must equal the plain pipeline's, bit for bit.
    import torch
    from pir_amd import distributed as D
    from gpu_helpers import all_to_all_in_process
    G = 2
    cuts = D.slot_cuts(orc.k * N, G)
    bufs = None
    for g in range(G):
        dbg = pir_amd.PIRDatabase.Create(pp, raw, slots=(cuts[g], cuts[g + 1]))
        dbg.finalize(release_staging=True)
        sg = pir_amd.PIRServer(dbg, pp)
        sg.set_galois_keys(keys)
        sg.set_concurrency(8)
        if g == 0:
            assert D.slots_exchange_supported(sg) and sg.scan_bytes() * G == info_bytes_whole(sg, dims, orc.k * N)
            bufs = [D.SlotsBuffers(sg, G, r, G, torch, "cuda:0") for r in range(G)]
            sg.stage_batch(queries)
            for r in range(G):           # rank r expands query r
                sg.slots_expand_async(r, 1, bufs[r].packed_send.data_ptr(), bufs[r].sv.data_ptr(), cuts)
            sg.sync()
            all_to_all_in_process([b.packed_recv for b in bufs], [b.packed_send for b in bufs], [b.x1_recv for b in bufs],
                                  [b.x1_send for b in bufs])
        sg.slots_scan_async(bufs[g].packed_recv.data_ptr(), G, 1, bufs[g].rows_send.data_ptr())
        sg.sync()
        if g == G - 1:
            all_to_all_in_process([b.rows_recv for b in bufs], [b.rows_send for b in bufs], [b.x2_recv for b in bufs],
                                  [b.x2_send for b in bufs])
            for r in range(G):
                sg.slots_finish_async(bufs[r].rows_recv.data_ptr(), 1, bufs[r].sv.data_ptr(), cuts, bufs[r].replies.data_ptr())
            sg.sync()
        dbg.close()
    for r in range(G):
        assert np.array_equal(bufs[r].replies.cpu().numpy().view(np.uint64)[0], batch[r]), r


def info_bytes_whole(srv, dims, kN):
    """Bytes of the whole database in the scan's operand layout, from the geometry a context reports."""
    i = srv.scan_info()
    per_residue = i["digits"] - (0.5 if i["top_digit_nibble"] else 0.0)
    tiles = ((dims[0] + 15) // 16) * ((dims[1] + 15) // 16)
    return int(kN * tiles * 256 * per_residue)


def test_cfg4_full_size():
    m = oracle.BFV_DEFAULT[8192]
    _large_d2(8192, m[:3] + [m[4]], 1 << 22, 1024, [428, 427], 6, 48)


def test_cfg5_full_size():
    m = oracle.BFV_DEFAULT[16384]
    _large_d2(16384, m[:4] + [m[8]], 1 << 24, 288, [321, 321], 7, 59)
