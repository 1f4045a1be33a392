"""Shared helpers for PIR end-to-end tests (mirror of the reference's PIRTestingBase,
test_base.cpp:27-84, with numpy's PRNG in place of SEAL's)."""
import numpy as np

import oracle
from oracle.client import Client


def generate_test_db(db_size, elem_size, seed=42):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(db_size, elem_size), dtype=np.uint8)


class PirSetup:
    """SetUpParams + GenerateDB + client (test_base.cpp:39-84)."""

    def __init__(self, dbsize, elem_size=0, dimensions=1, N=4096, plain_bits=24, bits_per_coeff=0, seed=42,
                 moduli=None, t=None, client_seed=99):
        self.params = oracle.create_pir_parameters(dbsize, elem_size, dimensions, N=N, plain_bits=plain_bits,
                                                   bits_per_coeff_=bits_per_coeff, moduli=moduli, t=t)
        p = self.params
        self.orc = oracle.Oracle.from_params(p)
        self.raw = generate_test_db(dbsize, p.bytes_per_item, seed)
        rc, self.db_ntt = self.orc.db_encode(self.raw.tobytes(), dbsize, p.bytes_per_item, p.items_per_plaintext,
                                             p.eff_bits_per_coeff, p.num_pt)
        assert rc == 0
        self.client = Client(self.orc, seed=client_seed)
        self.galois_keys = self.client.galois_keys()

    def item(self, i):
        return self.raw[i].tobytes()


def oracle_partial_reply(orc, db_ntt, dims, lo, hi, sv_coeff):
    """Partial reply of the row shard [lo, hi) of dimension 0, computed by the oracle on those rows ONLY (cost
    proportional to the shard): db_ntt = the shard's plaintexts (row-major, local), sv_coeff = the full selection
    vector in coefficient form [dim_sum, 2, k, N].  The recursion of database.cpp:170-258 over dimension 0 is a sum
    over its indices, so restricting it to [lo, hi) with the matching selectors gives that shard's summand."""
    dims = list(dims)
    sub_dims = [hi - lo] + dims[1:]
    sv = np.concatenate([sv_coeff[lo:hi], sv_coeff[dims[0]:]]).copy()
    rc, part = orc.db_multiply(np.ascontiguousarray(db_ntt), sub_dims, sv)
    return rc, part
