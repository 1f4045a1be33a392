"""GPU parity of the digit-sliced int8-MFMA database scan (pir_amd/csrc/scan_mfma.hip) across its
geometry variants -- digits per residue (5 / 6 / 7), k-steps per chunk, column chunks, ragged row and
column tiles -- and of the batch pipeline built on it (batched expansion, groups of up to 8 queries,
two lanes).  Every reply is compared bit for bit with the CPU oracle; `scan_info` proves which path ran."""
import os

import numpy as np
import pytest

import oracle
import pir_amd
from gpu_helpers import to_product_params
from pir_fixtures import PirSetup

pytestmark = pytest.mark.gpu


def make(setup, shard=None):
    pp = to_product_params(setup.params)
    db = pir_amd.PIRDatabase.Create(pp, shard=shard)
    db.populate(setup.raw)
    srv = pir_amd.PIRServer(db, pp)
    srv.set_galois_keys(setup.galois_keys)
    return db, srv


def setup_with_dims(short, elem, dims, **kw):
    """PirSetup with an explicit (non-square) dimension vector and prod(dims) plaintexts minus `short`
    items: the reference accepts any dimensions whose product covers num_pt (database.cpp:140-168 only
    uses them as given)."""
    probe = oracle.create_pir_parameters(10, elem, 1, **{k: v for k, v in kw.items() if k in
                                                          ("N", "plain_bits", "moduli", "t")})
    pts = int(np.prod(dims))
    s = PirSetup(pts * probe.items_per_plaintext - short, elem, len(dims), **kw)
    assert s.params.num_pt == pts
    s.params.dimensions = list(dims)
    return s


def check_queries(s, srv, indexes, decode=True):
    for idx in indexes:
        q = s.client.create_query_for(s.params, idx)
        rc, exp = s.orc.process_query(s.db_ntt, s.params.dimensions, q, s.galois_keys)
        assert rc == 0
        got = srv.process_query(q)
        assert np.array_equal(got, exp), idx
        if decode:
            assert s.client.process_response(s.params, idx, got) == s.item(idx)


# (label, setup kwargs, expected digits, expected chunks, expected ksteps, PIRGPU_SCAN_MFMA_WIDE or None)
GEOMETRIES = [
    # N=4096, 36-bit primes -> 5 digits
    ("L5 17x70 (2 k-steps, ragged tiles)", dict(dbsize=0, elem=2048, dims=[17, 70], N=4096, plain_bits=24), 5, 1, 2, None),
    ("L5 33x9 (1 k-step)", dict(dbsize=0, elem=2048, dims=[33, 9], N=4096, plain_bits=24), 5, 1, 1, None),
    # 13 column groups = 4 k-steps: more than two waves per SIMD hold -> the 4-wave kernel, one chunk
    ("L5 9x200 (wide kernel, 4 k-steps)", dict(dbsize=3, elem=2048, dims=[9, 200], N=4096, plain_bits=24), 5, 1, 4, None),
    # the same matrix on the 8-wave kernel: 2 equal chunks of 7 and 6 groups = 2 k-steps each
    ("L5 9x200 (8-wave kernel, 2 column chunks)", dict(dbsize=3, elem=2048, dims=[9, 200], N=4096, plain_bits=24), 5, 2, 2, "0"),
    # narrow matrix forced onto the 4-wave kernel (3 k-steps is its smallest instantiation)
    ("L5 17x150 (wide kernel forced, 3 k-steps)", dict(dbsize=4, elem=2048, dims=[17, 150], N=4096, plain_bits=24), 5, 1, 3, "1"),
    # 31 column groups = 8 k-steps: wider than the 7 the 4-wave kernel holds -> 2 chunks of 16 / 15 groups, 4 k-steps each
    ("L5 9x490 (wide kernel, 2 chunks of 4 k-steps)", dict(dbsize=1, elem=2048, dims=[9, 490], N=4096, plain_bits=24), 5, 2, 4, None),
    ("L5 d=3 4x4x40 (rows = 16)", dict(dbsize=1, elem=2048, dims=[4, 4, 40], N=4096, plain_bits=20), 5, 1, 1, None),
]


@pytest.mark.parametrize("top4", ["1", "0"], ids=["top digit nibble", "top digit byte"])
@pytest.mark.parametrize("label,kw,digits,chunks,ksteps,wide", GEOMETRIES, ids=[g[0] for g in GEOMETRIES])
def test_mfma_scan_geometries(label, kw, digits, chunks, ksteps, wide, top4, monkeypatch):
    kw = dict(kw)
    if wide is not None:
        monkeypatch.setenv("PIRGPU_SCAN_MFMA_WIDE", wide)
    monkeypatch.setenv("PIRGPU_SCAN_MFMA_TOP4", top4)
    s = setup_with_dims(kw.pop("dbsize"), kw.pop("elem"), kw.pop("dims"), **kw)
    db, srv = make(s)
    info = srv.scan_info()
    assert info["mfma"] and info["digits"] == digits and info["chunks"] == chunks and info["ksteps"] == ksteps, info
    assert info["top_digit_nibble"] == (top4 == "1"), info      # 36-bit moduli: the nibble form is the default
    n = s.params.num_items
    check_queries(s, srv, [0, n // 2 + 1, n - 1])
    # the same through the batch pipeline (one group of 3)
    idx = [1, n // 3, n - 2]
    queries = np.stack([s.client.create_query_for(s.params, i) for i in idx])
    got = srv.process_batch(queries, n_workers=4)
    for i, j in enumerate(idx):
        rc, exp = s.orc.process_query(s.db_ntt, s.params.dimensions, queries[i], s.galois_keys)
        assert np.array_equal(got[i], exp)
    db.close()


def test_single_query_through_chunked_mfma_scan(monkeypatch):
    """Matrices wider than one chunk scan single queries with the 64-bit kernels by default; forced onto the
    MFMA scan they fold the per-chunk partial sums (reduce_splits_kernel) to the same bits."""
    monkeypatch.setenv("PIRGPU_SCAN_MFMA_WIDE", "0")   # 8-wave kernel: 13 column groups = 2 chunks
    s = setup_with_dims(3, 2048, [9, 200], N=4096, plain_bits=24)
    db, srv = make(s)
    assert srv.scan_info()["mfma"] and not srv.scan_info()["single_query_mfma"]
    check_queries(s, srv, [11])
    db.close()
    monkeypatch.setenv("PIRGPU_SCAN_MFMA_SINGLE", "1")
    db, srv = make(s)
    assert srv.scan_info()["single_query_mfma"] and srv.scan_info()["chunks"] == 2
    check_queries(s, srv, [11, s.params.num_items - 1])
    db.close()


def test_mfma_scan_six_digits_n8192():
    # BFVDefault(8192): 43-bit data primes -> 6 digits, 2 k-steps per chunk
    m = oracle.BFV_DEFAULT[8192]
    s = setup_with_dims(2, 1024, [9, 130], N=8192, moduli=m[:3] + [m[4]],
                        t=oracle.plain_modulus_batching(8192, 24))
    db, srv = make(s)
    info = srv.scan_info()
    assert info["mfma"] and info["digits"] == 6 and info["ksteps"] == 3 and info["chunks"] == 1, info
    check_queries(s, srv, [5, s.params.num_items - 1])
    db.close()


def test_mfma_scan_six_digits_wide_kernel_n8192():
    # 17 column groups = 5 k-steps of 6 digits: the 4-wave kernel (one wave per SIMD, operands in VGPRs + AGPRs)
    m = oracle.BFV_DEFAULT[8192]
    s = setup_with_dims(2, 1024, [9, 270], N=8192, moduli=m[:3] + [m[4]],
                        t=oracle.plain_modulus_batching(8192, 24))
    db, srv = make(s)
    info = srv.scan_info()
    assert info["mfma"] and info["digits"] == 6 and info["ksteps"] == 5 and info["chunks"] == 1, info
    check_queries(s, srv, [5, s.params.num_items - 1])
    db.close()


def test_mfma_scan_seven_digits_n16384():
    # BFVDefault(16384): 48/49-bit data primes -> 7 digits
    m = oracle.BFV_DEFAULT[16384]
    s = setup_with_dims(0, 288, [9, 10], N=16384, moduli=m[:4] + [m[8]],
                        t=oracle.plain_modulus_batching(16384, 24))
    db, srv = make(s)
    info = srv.scan_info()
    assert info["mfma"] and info["digits"] == 7, info
    check_queries(s, srv, [163 * 45 + 7], decode=False)
    db.close()


def test_valu_scan_still_selectable(monkeypatch):
    """PIRGPU_SCAN_MFMA=0 keeps the 64-bit multiply-accumulate kernels (read when the context is created)."""
    monkeypatch.setenv("PIRGPU_SCAN_MFMA", "0")
    s = setup_with_dims(0, 2048, [17, 17], N=4096, plain_bits=24)
    db, srv = make(s)
    assert not srv.scan_info()["mfma"]
    check_queries(s, srv, [3, s.params.num_items - 1])
    db.close()


def test_options_by_name_replace_the_environment():
    """pirgpu_set_option: the same switches without touching the environment; workspace-shaping options are refused
    once the context has been used, unknown names are InvalidArgument."""
    s = setup_with_dims(0, 2048, [17, 17], N=4096, plain_bits=24)
    pp = to_product_params(s.params)
    db = pir_amd.PIRDatabase.Create(pp)
    assert db.get_option("scan_mfma") == -1                    # built-in default
    db.set_option("scan_mfma", 0)
    db.set_option("Head_Levels", 3)                            # case-insensitive
    with pytest.raises(pir_amd.PirGpuError) as e:              # settled sweeps are constants now (round 5), not options
        db.set_option("lanes", 1)
    assert e.value.code == pir_amd.StatusCode.INVALID_ARGUMENT
    assert db.get_option("SCAN_MFMA") == 0
    db.populate(s.raw)
    srv = pir_amd.PIRServer(db, pp)
    srv.set_galois_keys(s.galois_keys)
    assert not srv.scan_info()["mfma"]
    check_queries(s, srv, [3])
    with pytest.raises(pir_amd.PirGpuError) as e:
        db.set_option("scan_mfma", 1)
    assert e.value.code == pir_amd.StatusCode.FAILED_PRECONDITION
    with pytest.raises(pir_amd.PirGpuError) as e:
        db.set_option("no_such_option", 1)
    assert e.value.code == pir_amd.StatusCode.INVALID_ARGUMENT
    db.set_option("scan_mfma_wgs_batch", 96)                   # a run-time option: accepted at any time
    db.close()


def test_small_row_counts_use_the_valu_scan():
    s = PirSetup(300, 2048, 2, N=4096, plain_bits=24)        # 60 plaintexts -> dims [8, 8]: rows = 8 -> MFMA
    db, srv = make(s)
    assert srv.scan_info()["rows"] == 8 and srv.scan_info()["mfma"]
    db.close()
    s = PirSetup(100, 2048, 2, N=4096, plain_bits=24)        # 20 plaintexts -> dims [5, 4]: too few rows
    db, srv = make(s)
    assert not srv.scan_info()["mfma"]
    check_queries(s, srv, [99])
    db.close()


@pytest.mark.parametrize("count,workers", [(11, 16), (19, 16), (9, 3), (8, 8)])
def test_batch_groups_and_lanes(count, workers):
    """Ragged groups (8 + 3), several rounds and both lanes: every reply equals the single-query reply."""
    s = setup_with_dims(4, 2048, [20, 21], N=4096, plain_bits=24)
    db, srv = make(s)
    n = s.params.num_items
    idx = [(97 * i + 13) % n for i in range(count)]
    queries = np.stack([s.client.create_query_for(s.params, i) for i in idx])
    got = srv.process_batch(queries, n_workers=workers)
    assert got.shape[0] == count
    for i in (0, count // 2, count - 1):
        rc, exp = s.orc.process_query(s.db_ntt, s.params.dimensions, queries[i], s.galois_keys)
        assert np.array_equal(got[i], exp)
    for i in range(count):
        assert np.array_equal(got[i], srv.process_query(queries[i])), i
    # a second batch on the same context (lanes and workers reused)
    got2 = srv.process_batch(queries[::-1].copy(), n_workers=workers)
    assert np.array_equal(got2[::-1], got)
    db.close()


def test_repopulate_after_packing():
    """Loading new contents after the first query re-packs the operand-layout copy."""
    s1 = setup_with_dims(0, 2048, [16, 16], N=4096, plain_bits=24, seed=1)
    s2 = setup_with_dims(0, 2048, [16, 16], N=4096, plain_bits=24, seed=2)
    db, srv = make(s1)
    check_queries(s1, srv, [77])
    db.populate(s2.raw)
    srv.set_galois_keys(s2.galois_keys)
    check_queries(s2, srv, [77, 1000])
    assert np.array_equal(db.read_plaintext(5), s2.db_ntt[5])
    db.close()


def test_row_shards_with_mfma_scan():
    """Row shards of 24 + 24 rows, each MFMA-scanned; partial replies add up to the full reply (mod q)."""
    s = setup_with_dims(0, 2048, [48, 10], N=4096, plain_bits=24)
    q = s.client.create_query_for(s.params, 1234)
    rc, exp = s.orc.process_query(s.db_ntt, s.params.dimensions, q, s.galois_keys)
    acc = None
    for shard in ((0, 24), (24, 48)):
        db, srv = make(s, shard=shard)
        assert srv.scan_info()["mfma"] and srv.scan_info()["rows"] == 24
        part = srv.process_query(q)
        acc = part.copy() if acc is None else acc + part
        db.close()
    for j, qj in enumerate(s.orc.moduli[: s.orc.k]):
        acc[:, :, j, :] %= np.uint64(qj)
    assert np.array_equal(acc, exp)


def test_query_parallel_ranks_with_mfma_scan():
    """The multi-GPU step simulated on one GPU with MFMA-scanned shards: each 'rank' expands its block of the
    batch into the shared buffer, every rank runs ALL queries from the gathered selection vectors
    (pirgpu_batch_run_selectors -> the ext_sv path of the batch pipeline), partial replies add up mod q."""
    import torch
    from pir_amd.distributed import owned_queries
    s = setup_with_dims(0, 2048, [48, 10], N=4096, plain_bits=24)
    p = s.params
    count, world = 6, 2
    idx = [(211 * i + 17) % p.num_items for i in range(count)]
    queries = np.stack([s.client.create_query_for(p, i) for i in idx])
    sv_all = torch.zeros((count, p.dim_sum, 2, s.orc.k, 4096), dtype=torch.int64, device="cuda")
    ranks = []
    for r, shard in enumerate(((0, 24), (24, 48))):
        db, srv = make(s, shard=shard)
        assert srv.scan_info()["mfma"]
        srv.set_concurrency(3)
        srv.stage_batch(queries)
        lo, hi = owned_queries(count, r, world)
        srv.batch_expand(lo, hi - lo, sv_all[lo].data_ptr())
        ranks.append((db, srv))
    torch.cuda.synchronize()
    acc = None
    for db, srv in ranks:
        srv.batch_run_selectors(sv_all.data_ptr(), count)
        part = srv.fetch_batch()
        acc = part.copy() if acc is None else acc + part
    for j, qj in enumerate(s.orc.moduli[: s.orc.k]):
        acc[:, :, :, j, :] %= np.uint64(qj)
    for i in range(count):
        rc, exp = s.orc.process_query(s.db_ntt, p.dimensions, queries[i], s.galois_keys)
        assert rc == 0 and np.array_equal(acc[i], exp), i
    for db, srv in ranks:
        db.close()


def test_finalize_and_release_staging():
    """pirgpu_db_finalize(release_staging): only the operand-layout copy stays resident; queries (single and
    batch) and read_plaintext keep working bit-exactly, reloading is refused."""
    s = setup_with_dims(2, 2048, [17, 19], N=4096, plain_bits=24)
    db, srv = make(s)
    db.finalize(release_staging=True)
    assert srv.scan_info()["single_query_mfma"]
    check_queries(s, srv, [0, s.params.num_items - 1])
    for i in (0, 18, 19, s.params.num_pt - 1):
        assert np.array_equal(db.read_plaintext(i), s.db_ntt[i]), i
    idx = [3, 500, 1000]
    queries = np.stack([s.client.create_query_for(s.params, i) for i in idx])
    got = srv.process_batch(queries, n_workers=4)
    for i in range(len(idx)):
        rc, exp = s.orc.process_query(s.db_ntt, s.params.dimensions, queries[i], s.galois_keys)
        assert np.array_equal(got[i], exp)
    with pytest.raises(pir_amd.PirGpuError) as e:
        db.populate(s.raw)
    assert e.value.code == pir_amd.StatusCode.FAILED_PRECONDITION
    db.close()
    # chunked geometry: releasing the staging copy moves single queries onto the MFMA scan as well
    s = setup_with_dims(1, 2048, [9, 490], N=4096, plain_bits=24)
    db, srv = make(s)
    assert not srv.scan_info()["single_query_mfma"]
    db.finalize(release_staging=True)
    assert srv.scan_info()["single_query_mfma"]
    check_queries(s, srv, [77])
    db.close()


def test_mfma_and_valu_scan_agree_on_boundary_selectors(monkeypatch):
    """Selection vectors made of the boundary residues of the centring / digit decomposition (0, 1, q-1, q/2,
    q/2 +- 1, 0x7F / 0x80 byte patterns, and the limits of the asymmetric centring that lets the top digit fit a nibble:
    vmax = 7 * 256^4 + 127 (256^4 - 1) / 255 and its neighbours, 2^35 -+ ...), injected in NTT form through
    pirgpu_batch_run_selectors: the int8-MFMA scan with the top digit as a nibble, the same with full bytes and the
    64-bit multiply-accumulate scan must produce identical replies."""
    import torch
    s = setup_with_dims(1, 2048, [17, 19], N=4096, plain_bits=24)
    p = s.params
    k, N = s.orc.k, 4096
    count = 5
    rng = np.random.default_rng(123)
    sv = np.empty((count, p.dim_sum, 2, k, N), dtype=np.uint64)
    for j in range(k):
        q = int(s.orc.moduli[j])
        half = q >> 1
        vmax = 7 * 256 ** 4 + 127 * ((256 ** 4 - 1) // 255)      # largest value whose top digit is 7 (scan_mfma.hip)
        cases = np.array([0, 1, q - 1, q - 2, half, half + 1, half - 1, half + 2, 0x7F, 0x80, 0x81,
                          0x7F7F7F7F7F % q, 0x8080808080 % q, 0x807F807F80 % q, q - 0x80, q - 0x8080,
                          vmax, vmax + 1, vmax + 2, vmax - 1, 2 ** 35, 2 ** 35 - 1, 2 ** 35 - 2155905152,
                          2 ** 35 - 2155905153, 7 * 256 ** 4, 7 * 256 ** 4 - 1, 0x0F80808080 % q], dtype=np.uint64)
        assert cases.max() < q
        pick = rng.integers(0, len(cases), size=(count, p.dim_sum, 2, N))
        sv[:, :, :, j, :] = cases[pick]
    sv_dev = torch.from_numpy(sv.view(np.int64)).cuda()
    replies = []
    for mfma, top4 in (("1", "1"), ("1", "0"), ("0", "1")):
        monkeypatch.setenv("PIRGPU_SCAN_MFMA", mfma)
        monkeypatch.setenv("PIRGPU_SCAN_MFMA_TOP4", top4)
        db, srv = make(s)
        assert srv.scan_info()["mfma"] == (mfma == "1")
        assert srv.scan_info()["top_digit_nibble"] == (mfma == "1" and top4 == "1")
        srv.set_concurrency(8)
        srv.stage_batch(np.zeros((count, 1, 2, k, N), dtype=np.uint64))   # sizes the reply buffers
        srv.batch_run_selectors(sv_dev.data_ptr(), count)
        replies.append(srv.fetch_batch())
        db.close()
    assert np.array_equal(replies[0], replies[1]) and np.array_equal(replies[0], replies[2])
    assert replies[0].any()
