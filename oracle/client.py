"""CPU BFV client used to drive the oracle and the GPU path -- TEST INFRASTRUCTURE ONLY.

Restates the client side of the reference (``client.cpp``) plus the SEAL 3.5.6
key generation / encryption / decryption it calls, so that tests can create
real queries and check replies at plaintext level:

* ``createQueryFor``          -- client.cpp:92-144
* ``ProcessReplyCiphertextDecomp`` -- client.cpp:219-255
* ``ProcessResponse`` (string decode) -- client.cpp:160-185
* KeyGenerator / Encryptor / Decryptor -- SEAL 3.5.6 semantics (RLWE public-key
  encryption at key level followed by divide-and-round by the special prime,
  Galois keys as one RLWE sample per RNS digit carrying p * sigma_g(s)).

Randomness comes from numpy's PCG64 with a caller-supplied seed (SEAL's Blake2
PRNG is not available); the server path is deterministic given its inputs, so
this does not affect parity.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

from . import (Oracle, PirParams, calculate_indices, calculate_item_offset, generate_galois_elts,
               next_power_two, string_decode)

NOISE_SIGMA = 3.2            # SEAL default noise standard deviation
NOISE_MAX_DEV = 6 * NOISE_SIGMA


class Client:
    def __init__(self, orc: Oracle, seed: int = 1):
        self.o = orc
        self.N, self.k, self.t = orc.N, orc.k, orc.t
        self.q = [int(x) for x in orc.moduli]          # k data + special
        self.rng = np.random.default_rng(seed)
        self.Q = 1
        for x in self.q[: self.k]:
            self.Q *= x
        self._keygen()

    # -- sampling -------------------------------------------------------------
    def _to_rns(self, signed: np.ndarray, nmod: int) -> np.ndarray:
        """signed int64 coefficients -> [nmod, N] canonical residues."""
        out = np.empty((nmod, self.N), dtype=np.uint64)
        for i in range(nmod):
            q = self.q[i]
            out[i] = np.where(signed < 0, (signed + q), signed).astype(np.uint64)
        return out

    def _ternary(self):
        return self.rng.integers(-1, 2, size=self.N, dtype=np.int64)

    def _noise(self):
        e = np.rint(self.rng.normal(0.0, NOISE_SIGMA, size=self.N))
        e = np.clip(e, -NOISE_MAX_DEV, NOISE_MAX_DEV)
        return e.astype(np.int64)

    def _uniform(self, nmod):
        out = np.empty((nmod, self.N), dtype=np.uint64)
        for i in range(nmod):
            out[i] = self.rng.integers(0, self.q[i], size=self.N, dtype=np.uint64)
        return out

    def _ntt_all(self, x):
        return np.stack([self.o.ntt_fwd(i, x[i]) for i in range(x.shape[0])])

    def _intt_all(self, x):
        return np.stack([self.o.ntt_inv(i, x[i]) for i in range(x.shape[0])])

    # -- keys -------------------------------------------------------------------
    def _keygen(self):
        km = self.k + 1
        self.s_signed = self._ternary()
        self.s_ntt = self._ntt_all(self._to_rns(self.s_signed, km))     # [k+1, N]
        a = self._uniform(km)                                            # NTT form
        e = self._ntt_all(self._to_rns(self._noise(), km))
        pk0 = np.empty_like(a)
        for i in range(km):
            as_e = self.o.poly_add(i, self.o.dyadic_mul(i, a[i], self.s_ntt[i]), e[i])
            pk0[i] = self.o.poly_neg(i, as_e)
        self.pk = (pk0, a)

    def _rlwe_zero_sym(self):
        """(-(a s + e), a) at key level, NTT form."""
        km = self.k + 1
        a = self._uniform(km)
        e = self._ntt_all(self._to_rns(self._noise(), km))
        c0 = np.empty_like(a)
        for i in range(km):
            c0[i] = self.o.poly_neg(i, self.o.poly_add(i, self.o.dyadic_mul(i, a[i], self.s_ntt[i]), e[i]))
        return c0, a

    def galois_key(self, g: int) -> np.ndarray:
        """KSwitchKey for sigma_g(s): ndarray [k, 2, k+1, N], NTT form."""
        km, k = self.k + 1, self.k
        s_rns = self._to_rns(self.s_signed, km)
        new_key = np.stack([self.o.ntt_fwd(i, self.o.apply_galois_poly(i, s_rns[i], g)) for i in range(km)])
        key = np.empty((k, 2, km, self.N), dtype=np.uint64)
        p = self.q[k]
        for j in range(k):
            c0, c1 = self._rlwe_zero_sym()
            factor = np.full(self.N, p % self.q[j], dtype=np.uint64)
            c0[j] = self.o.poly_add(j, c0[j], self.o.dyadic_mul(j, new_key[j], factor))
            key[j, 0], key[j, 1] = c0, c1
        return key

    def galois_keys(self, elts: Optional[Sequence[int]] = None) -> Dict[int, np.ndarray]:
        if elts is None:
            elts = generate_galois_elts(self.N)                       # client.cpp:47
        return {g: self.galois_key(g) for g in elts}

    # -- encrypt / decrypt ------------------------------------------------------
    def encrypt(self, coeffs) -> np.ndarray:
        """Public-key BFV encryption of a plaintext given as coefficients < t. -> [2, k, N]."""
        km, k, t = self.k + 1, self.k, self.t
        m = np.zeros(self.N, dtype=np.uint64)
        coeffs = np.asarray(coeffs, dtype=np.uint64)
        m[: coeffs.shape[0]] = coeffs
        u = self._ntt_all(self._to_rns(self._ternary(), km))
        ct = np.empty((2, k, self.N), dtype=np.uint64)
        for comp in range(2):
            x = np.stack([self.o.dyadic_mul(i, self.pk[comp][i], u[i]) for i in range(km)])
            x = self._intt_all(x)
            e = self._to_rns(self._noise(), km)
            for i in range(km):
                x[i] = self.o.poly_add(i, x[i], e[i])
            ct[comp] = self.o.divide_round_special(x)
        # multiply_add_plain_with_scaling_variant: round(Q m / t) added to c0
        delta = self.Q // t
        q_mod_t = self.Q % t
        half = (t + 1) >> 1
        nz = np.nonzero(m)[0]
        for idx in nz:
            mv = int(m[idx])
            fix = (mv * q_mod_t + half) // t
            for j in range(k):
                qj = self.q[j]
                add = (mv * (delta % qj) + fix) % qj
                ct[0, j, idx] = (int(ct[0, j, idx]) + add) % qj
        return ct

    def _phase(self, ct) -> List[int]:
        """[c0 + c1 s]_Q as python ints (centered not applied)."""
        k = self.k
        res = []
        for j in range(k):
            c1s = self.o.ntt_inv(j, self.o.dyadic_mul(j, self.o.ntt_fwd(j, ct[1, j]), self.s_ntt[j]))
            res.append(self.o.poly_add(j, ct[0, j], c1s))
        # CRT compose
        Q = self.Q
        xs = [0] * self.N
        for j in range(k):
            qj = self.q[j]
            Mj = Q // qj
            inv = pow(Mj % qj, -1, qj)
            f = (Mj * inv) % Q
            col = res[j].tolist()
            for i in range(self.N):
                xs[i] = (xs[i] + col[i] * f) % Q
        return xs

    def decrypt(self, ct) -> np.ndarray:
        """-> plaintext coefficients [N] in [0, t)."""
        t, Q = self.t, self.Q
        xs = self._phase(np.asarray(ct))
        return np.array([((t * x + (Q >> 1)) // Q) % t for x in xs], dtype=np.uint64)

    def noise_budget(self, ct) -> float:
        """Invariant noise budget in bits (SEAL Decryptor::invariant_noise_budget)."""
        import math
        t, Q = self.t, self.Q
        worst = 0
        for x in self._phase(np.asarray(ct)):
            r = (t * x) % Q
            if r > Q // 2:
                r = Q - r
            worst = max(worst, r)
        if worst == 0:
            return float(Q.bit_length())
        return max(0.0, math.log2(Q) - math.log2(worst) - 1)

    # -- PIR client (client.cpp) -------------------------------------------------
    def create_query_for(self, params: PirParams, desired_index: int) -> np.ndarray:
        """client.cpp:92-144 -> [num_query_cts, 2, k, N]."""
        if desired_index >= params.num_items:
            raise ValueError("invalid index %d" % desired_index)
        N, t = self.N, self.t
        dims = list(params.dimensions)
        indices = calculate_indices(desired_index, params.items_per_plaintext, dims)
        dim_sum = params.dim_sum
        offset = 0
        nq = dim_sum // N + 1
        out = np.empty((nq, 2, self.k, N), dtype=np.uint64)
        for c in range(nq):
            pt = np.zeros(N, dtype=np.uint64)
            while indices:
                if indices[0] + offset >= N:
                    indices[0] -= (N - offset)
                    dims[0] -= (N - offset)
                    offset = 0
                    break
                m = N if c < nq - 1 else next_power_two(dim_sum % N)
                pt[indices[0] + offset] = pow(m, -1, t)
                offset += dims[0]
                indices.pop(0)
                dims.pop(0)
                if offset >= N:
                    offset -= N
                    break
            out[c] = self.encrypt(pt)
        return out

    def process_reply(self, params: PirParams, reply_cts: np.ndarray) -> np.ndarray:
        """ProcessReplyCiphertextDecomp (client.cpp:219-255) -> plaintext coefficients."""
        exp_ratio = self.o.expansion_ratio() * 2
        nd = len(params.dimensions)
        if reply_cts.shape[0] != exp_ratio ** (nd - 1):
            raise ValueError("Number of ciphertexts in reply does not match expected")
        cts = [np.asarray(c) for c in reply_cts]
        pts = []
        for _ in range(nd):
            pts = [self.decrypt(c) for c in cts]
            if len(pts) <= 1:
                break
            cts = [self.o.redecode(np.stack(pts[i * exp_ratio:(i + 1) * exp_ratio]))
                   for i in range(len(cts) // exp_ratio)]
        return pts[0]

    def process_response(self, params: PirParams, index: int, reply_cts: np.ndarray) -> bytes:
        """ProcessResponse (client.cpp:160-185) for one reply."""
        pt = self.process_reply(params, reply_cts)
        rc, data = string_decode(pt, params.eff_bits_per_coeff, params.bytes_per_item,
                                 calculate_item_offset(index, params.items_per_plaintext, params.bytes_per_item))
        if rc != 0:
            raise ValueError("Requested decode beyond end of data in polynomial")
        return data
