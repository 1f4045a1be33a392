"""Every kernel family under all three arithmetic flavours of the transforms (PIRGPU_NTT_MODE = 0 integer
Shoup/Harvey, 1 exact fp64, 2 wide fp64), each compared bit for bit with the CPU oracle: the flavours must be
indistinguishable in their residues.  Families: batched forward / inverse NTT, key switch (substitute), the
expansion tree (ks_digit / ks_mac_intt / ks_combine), database encode, scan + upper recursion level + batch path."""
import numpy as np
import pytest

import oracle
import pir_amd
from gpu_helpers import random_ct, random_key, to_product_params
from pir_fixtures import PirSetup

pytestmark = pytest.mark.gpu

CHAINS = {
    "n4096_36bit": (4096, oracle.BFV_DEFAULT[4096], 1),                                       # default flavour 1
    "n8192_44bit": (8192, oracle.BFV_DEFAULT[8192][:3] + [oracle.BFV_DEFAULT[8192][4]], 1),
    "n2048_27bit": (2048, oracle.coeff_modulus_create(2048, [27, 27]), 1),
    "n16384_49bit": (16384, oracle.BFV_DEFAULT[16384][:4] + [oracle.BFV_DEFAULT[16384][8]], 2),  # default flavour 2
}


def _server(s, load_db=True):
    pp = to_product_params(s.params)
    db = pir_amd.PIRDatabase.Create(pp)
    if load_db:
        db.populate(s.raw)
    return db, pir_amd.PIRServer(db, pp)


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("chain", sorted(CHAINS))
def test_transform_and_key_switch_flavours(monkeypatch, chain, mode):
    N, moduli, default = CHAINS[chain]
    monkeypatch.setenv("PIRGPU_NTT_MODE", str(mode))
    s = PirSetup(12, 0, 1, N=N, plain_bits=20, moduli=moduli)
    db, srv = _server(s)
    # a request for a narrower flavour than the moduli allow keeps the default one
    assert srv.ntt_mode() == (mode if mode in (0, 2) or mode == default else default)
    rng = np.random.default_rng(N + mode)
    cts = random_ct(s.orc, rng, 2)
    fwd = srv.ntt_forward(cts)
    assert np.array_equal(fwd, np.stack([s.orc.ct_ntt_fwd(c) for c in cts]))
    assert np.array_equal(srv.ntt_inverse(fwd), cts)
    for g in (3, N + 1, N // 4 + 1):
        key = random_key(s.orc, rng)
        srv.set_galois_keys({g: key})
        rc, exp = s.orc.apply_galois_ct(cts[0], g, key)
        assert rc == 0 and np.array_equal(srv.substitute_power_x_inplace(cts[0].copy(), g), exp)
    keys = {(N >> j) + 1: random_key(s.orc, rng) for j in range(4)}
    srv.set_galois_keys(keys)
    rc, exp = s.orc.oblivious_expansion(cts[1], 11, keys)
    assert rc == 0 and np.array_equal(srv.oblivious_expansion(cts[1], 11), exp)
    for i in range(s.params.num_pt):
        assert np.array_equal(db.read_plaintext(i), s.db_ntt[i])
    db.close()


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("chain,items,elem", [("n4096_36bit", 2600, 288), ("n8192_44bit", 1500, 1024),
                                              ("n16384_49bit", 11000, 288)])
def test_query_path_flavours(monkeypatch, chain, items, elem, mode):
    """d = 2 with >= 8 rows: MFMA scan, fused upper level (re-encode + lift + NTT + MAC), final inverse NTT, the
    batched expansion -- replies of single and batched queries equal the oracle's in every flavour."""
    N, moduli, default = CHAINS[chain]
    monkeypatch.setenv("PIRGPU_NTT_MODE", str(mode))
    s = PirSetup(items, elem, 2, N=N, plain_bits=24, moduli=moduli)
    db, srv = _server(s)
    assert srv.scan_info()["mfma"]
    rng = np.random.default_rng(mode)
    keys = {(N >> j) + 1: random_key(s.orc, rng) for j in range(N.bit_length() - 1)}
    srv.set_galois_keys(keys)
    queries = random_ct(s.orc, rng, 3)[:, None]          # [3 queries, 1 ct each, 2, k, N]
    rc, exp = s.orc.process_query(s.db_ntt, s.params.dimensions, queries[0], keys)
    assert rc == 0
    assert np.array_equal(srv.process_query(queries[0]), exp)
    batch = srv.process_batch(queries, n_workers=3)
    assert np.array_equal(batch[0], exp)
    for i in (1, 2):
        assert np.array_equal(batch[i], srv.process_query(queries[i]))
    db.close()


@pytest.mark.parametrize("d,items,elem,mb", [(2, 2600, 288, 1), (2, 2600, 288, 64), (3, 700, 288, 1)])
def test_split_upper_level_matches_fused(monkeypatch, d, items, elem, mb):
    """The upper recursion level as transform-to-scratch + elementwise multiply-accumulate (the N = 16384 default,
    where the fused kernel cannot hold its accumulators) forced on at N = 4096, with a scratch budget small enough
    for several blocks of children: single and batched replies equal the oracle's, for d = 2 and d = 3."""
    monkeypatch.setenv("PIRGPU_SPLIT_UPPER", "1")
    monkeypatch.setenv("PIRGPU_SPLIT_UPPER_MB", str(mb))
    s = PirSetup(items, elem, d, N=4096, plain_bits=24)
    db, srv = _server(s)
    rng = np.random.default_rng(d)
    keys = {(4096 >> j) + 1: random_key(s.orc, rng) for j in range(12)}
    srv.set_galois_keys(keys)
    queries = random_ct(s.orc, rng, 3)[:, None]
    exp = [s.orc.process_query(s.db_ntt, s.params.dimensions, q, keys)[1] for q in queries]
    for i in range(3):
        assert np.array_equal(srv.process_query(queries[i]), exp[i]), i
    batch = srv.process_batch(queries, n_workers=3)
    for i in range(3):
        assert np.array_equal(batch[i], exp[i]), i
    db.close()


@pytest.mark.parametrize("knob", ["PIRGPU_LAST_NTT", "PIRGPU_FUSE_LAST", "PIRGPU_FUSE_MAC_COMBINE", "PIRGPU_PACK40",
                                  "PIRGPU_SEL_F64", "PIRGPU_TREE40"])
def test_expansion_fallback_paths(monkeypatch, knob):
    """The expansion's optional fusions switched off one at a time (coefficient-domain last level, unfused last level,
    separate combine pass, digits as doubles, lane selectors as u64): every variant must give the oracle's reply, single and batched."""
    monkeypatch.setenv(knob, "0")
    N, moduli, _ = CHAINS["n4096_36bit"]
    s = PirSetup(5000, 288, 2, N=N, plain_bits=24, moduli=moduli)
    db, srv = _server(s)
    rng = np.random.default_rng(11)
    keys = {(N >> j) + 1: random_key(s.orc, rng) for j in range(N.bit_length() - 1)}
    srv.set_galois_keys(keys)
    queries = random_ct(s.orc, rng, 9)[:, None]
    rc, exp = s.orc.process_query(s.db_ntt, s.params.dimensions, queries[0], keys)
    assert rc == 0
    assert np.array_equal(srv.process_query(queries[0]), exp)
    batch = srv.process_batch(queries, n_workers=9)          # one full group of 8 + a group of 1
    assert np.array_equal(batch[0], exp)
    rc, exp8 = s.orc.process_query(s.db_ntt, s.params.dimensions, queries[8], keys)
    assert rc == 0 and np.array_equal(batch[8], exp8)
    db.close()


@pytest.mark.parametrize("chain,items,off,d", [("n4096_36bit", 45000, None, 2), ("n4096_36bit", 45000, "PIRGPU_TREE40", 2),
                                               ("n4096_36bit", 45000, "PIRGPU_PACK40", 2), ("n8192_44bit", 95000, None, 2),
                                               ("n16384_49bit", 49000, None, 2), ("n4096_36bit", 1400, None, 3),
                                               ("n4096_36bit", 45000, "PIRGPU_LOOP_TRANSFORMS", 2),   # the one-transform-per-workgroup forms at the same sizes
                                               ("n16384_49bit", 49000, "PIRGPU_LOOP_TRANSFORMS", 2),
                                               ("n16384_49bit", 5000, None, 3)])
def test_looped_transforms_match_the_oracle(monkeypatch, chain, items, off, d):
    """LOOP_TRANSFORMS (the default of the fp64 flavours): at the wide expansion levels -- from 1024 (tree ciphertext,
    digit) pairs per launch on -- one workgroup runs the k + 1 digit transforms of a source polynomial, and at the split
    upper level (the N = 16384 default, forced on for the smaller rings here) all chunks x target moduli of a child's
    source polynomial, instead of one transform per workgroup.  A full group of 8 queries with more than 64 leaves
    (32 at k = 4) reaches those levels; every storage form of the digits (5-byte tree and digits, 5-byte digits only,
    doubles) must give the oracle's reply, single and batched; d = 3 runs two split upper levels."""
    N, moduli, _ = CHAINS[chain]
    monkeypatch.setenv("PIRGPU_LOOP_TRANSFORMS", "1")
    monkeypatch.setenv("PIRGPU_SPLIT_UPPER", "1")
    if off:
        monkeypatch.setenv(off, "0")
    s = PirSetup(items, 288, d, N=N, plain_bits=24, moduli=moduli)
    assert d == 3 or sum(s.params.dimensions) > (32 if N == 16384 else 64), s.params.dimensions
    db, srv = _server(s)
    rng = np.random.default_rng(5)
    keys = {(N >> j) + 1: random_key(s.orc, rng) for j in range(N.bit_length() - 1)}
    srv.set_galois_keys(keys)
    queries = random_ct(s.orc, rng, 9)[:, None]
    batch = srv.process_batch(queries, n_workers=9)          # one full group of 8 + a group of 1
    for i in (0, 8):
        rc, exp = s.orc.process_query(s.db_ntt, s.params.dimensions, queries[i], keys)
        assert rc == 0 and np.array_equal(batch[i], exp), i
    assert np.array_equal(srv.process_query(queries[3]), batch[3])
    db.close()


@pytest.mark.parametrize("chain,items,knob,value", [
    # round 6: the tree's c0 polynomials in NTT form from the first fused level on -- one kernel for both components of a
    # level (1) or component 0 as a launch of its own (2); both measured slower than the coefficient-form tree and off by
    # default (profiles/r06_ab_c0_ntt_*.txt), both must give the oracle's bits
    ("n4096_36bit", 45000, "PIRGPU_C0_NTT", "1"), ("n4096_36bit", 45000, "PIRGPU_C0_NTT", "2"),
    ("n8192_44bit", 95000, "PIRGPU_C0_NTT", "2"),
    # packed key-switch intermediates wider than they need be (6 / 7 bytes for 36-bit moduli), the 44-bit chain in 7 bytes
    # and in doubles, the 49-bit chain in doubles and with the tree packed as well
    ("n4096_36bit", 45000, "PIRGPU_PACK_BYTES", "6"), ("n4096_36bit", 45000, "PIRGPU_PACK_BYTES", "7"),
    ("n8192_44bit", 95000, "PIRGPU_PACK_BYTES", "7"), ("n8192_44bit", 95000, "PIRGPU_PACK_BYTES", "8"),
    ("n16384_49bit", 49000, "PIRGPU_PACK_BYTES", "8"), ("n16384_49bit", 49000, "PIRGPU_TREE40_WIDE", "1"),
    # the batch launches' scan fold in integer arithmetic again
    ("n4096_36bit", 45000, "PIRGPU_SCAN_F64_FOLD_BATCH", "0")])
def test_round6_knobs_match_the_oracle(monkeypatch, chain, items, knob, value):
    """Every storage width of the packed intermediates and every form of the c0-in-NTT-form tree gives the oracle's reply,
    for a full group of 8 queries (the fused wide levels and the NTT-domain last level are only reached by a group) and a
    single query."""
    N, moduli, _ = CHAINS[chain]
    monkeypatch.setenv(knob, value)
    s = PirSetup(items, 288, 2, N=N, plain_bits=24, moduli=moduli)
    db, srv = _server(s)
    rng = np.random.default_rng(23)
    keys = {(N >> j) + 1: random_key(s.orc, rng) for j in range(N.bit_length() - 1)}
    srv.set_galois_keys(keys)
    queries = random_ct(s.orc, rng, 9)[:, None]
    batch = srv.process_batch(queries, n_workers=9)          # one full group of 8 + a group of 1
    for i in (0, 5, 8):
        rc, exp = s.orc.process_query(s.db_ntt, s.params.dimensions, queries[i], keys)
        assert rc == 0 and np.array_equal(batch[i], exp), i
    assert np.array_equal(srv.process_query(queries[3]), batch[3])
    db.close()


def test_upper_level_lds_twiddle_form_still_matches_the_oracle():
    """upper_fused_kernel's LDS-twiddle form (the default of rounds 2 - 5) is chosen once per process
    (PIRGPU_UPPER_LDS_TW=1); since round 6 the plain form with the source prefetch is the default.  The multiply / query
    parity tests (d = 2 and d = 3 against the oracle) are re-run in a child process with the old form selected."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PIRGPU_UPPER_LDS_TW="1", PIRGPU_ALLOW_ENV="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-q", "-m", "gpu", "-x",
                        "-k", "multiply_and_query or multi_ciphertext_query"], env=env, capture_output=True, text=True,
                       timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_environment_knobs_need_the_gate():
    """PIRGPU_NTT_MODE (and every other PIRGPU_* knob of the library) is read only when
    PIRGPU_ALLOW_ENV=1 is set as well -- a server's arithmetic flavour does not depend on stray environment variables."""
    import os
    import subprocess
    import sys
    code = ("import pir_amd, sys; sys.path.insert(0, 'tests'); "
            "from pir_amd import parameters as P; "
            "pp = P.create_pir_parameters(100, 288, 2, P.generate_encryption_params(4096, 24)); "
            "db = pir_amd.PIRDatabase.Create(pp); srv = pir_amd.PIRServer(db, pp); print('MODE', srv.ntt_mode())")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for gate in ("0", "1"):
        env = dict(os.environ, PIRGPU_NTT_MODE="0", PIRGPU_ALLOW_ENV=gate)
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[gate] = int(r.stdout.split("MODE")[1].split()[0])
    assert out == {"0": 1, "1": 0}          # closed: the flavour the moduli call for (fp64); open: the forced integer flavour
