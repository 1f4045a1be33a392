"""Independent big-integer model of the BFV server primitives the oracle restates -- TEST INFRASTRUCTURE.

Written from the mathematical definitions in SURVEY.md App. A.2-A.5 (SEAL 3.5.6 semantics), NOT from
oracle/pir_oracle.c: no butterflies, no Barrett/Shoup tricks, no lazy ranges, no RNS shortcuts where a
plain integer formulation exists.  Everything is Python `int` arithmetic at toy ring degrees (N = 16, 32),
so a silent mistake in the oracle's reductions or rounding (the places where "decrypts the same" would
hide a wrong bit) shows up as a residue mismatch in tests/test_keyswitch_model.py.

  * negacyclic NTT        = evaluation of the polynomial at psi^(2*bitrev(i)+1), psi the MINIMAL primitive
                            2N-th root (App. A.2), done as an O(N^2) sum;
  * ring product          = schoolbook convolution mod (x^N + 1, q);
  * key switching         = App. A.4, in two independent ways: (1) the per-modulus formula with the
                            explicit +floor(p/2) rounding, (2) CRT-lift the (k+1)-residue accumulator to
                            an integer X in [0, Q*p) and take floor((X + floor(p/2)) / p) mod q_i
                            (what "divide by the special prime and round" means);
  * apply_galois          = App. A.3 index map;  x^-k = negacyclic shift (App. A.5);
  * oblivious expansion   = reference server.cpp:105-146 restated with the pieces above.
"""
from __future__ import annotations

from typing import Dict, List, Sequence


def bitrev(i: int, bits: int) -> int:
    r = 0
    for _ in range(bits):
        r = (r << 1) | (i & 1)
        i >>= 1
    return r


def minimal_primitive_root(two_n: int, q: int) -> int:
    """Smallest psi in [1, q) whose multiplicative order is exactly two_n (two_n a power of two dividing q-1)."""
    assert (q - 1) % two_n == 0
    e = (q - 1) // two_n
    for x in range(2, 1000):
        cand = pow(x, e, q)                         # order divides 2N ...
        if pow(cand, two_n // 2, q) == q - 1:       # ... and cand^N = -1  =>  order == 2N exactly
            # the primitive 2N-th roots are exactly cand^odd: take the minimum
            g2 = cand * cand % q
            best = cur = cand
            for _ in range(two_n // 2 - 1):
                cur = cur * g2 % q
                best = min(best, cur)
            return best
    raise ValueError("no primitive root")


class Ring:
    """Z_q[x]/(x^N+1) for one prime q with SEAL's NTT ordering."""

    def __init__(self, N: int, q: int):
        self.N, self.q = N, q
        self.logN = N.bit_length() - 1
        self.psi = minimal_primitive_root(2 * N, q)
        self.points = [pow(self.psi, 2 * bitrev(i, self.logN) + 1, q) for i in range(N)]

    def ntt(self, a: Sequence[int]) -> List[int]:
        q = self.q
        return [sum(int(c) * pow(x, j, q) for j, c in enumerate(a)) % q for x in self.points]

    def intt(self, v: Sequence[int]) -> List[int]:
        """Inverse by Lagrange on the 2N-th roots: a_j = N^-1 * sum_i v_i * x_i^-j."""
        q, N = self.q, self.N
        ninv = pow(N, -1, q)
        inv_pts = [pow(x, -1, q) for x in self.points]
        return [ninv * sum(int(v[i]) * pow(inv_pts[i], j, q) for i in range(N)) % q for j in range(N)]

    def mul(self, a: Sequence[int], b: Sequence[int]) -> List[int]:
        N, q = self.N, self.q
        out = [0] * N
        for i, x in enumerate(a):
            x = int(x)
            if not x:
                continue
            for j, y in enumerate(b):
                k = i + j
                if k < N:
                    out[k] += x * int(y)
                else:
                    out[k - N] -= x * int(y)
        return [v % q for v in out]


def apply_galois(a: Sequence[int], g: int, N: int, q: int) -> List[int]:
    """p(x) -> p(x^g) in Z_q[x]/(x^N+1): x^i -> x^(i*g mod 2N), x^N = -1."""
    out = [0] * N
    for i, c in enumerate(a):
        e = (i * g) % (2 * N)
        out[e % N] = (-int(c)) % q if e >= N else int(c) % q
    return out


def mul_monomial(a: Sequence[int], e: int, N: int, q: int) -> List[int]:
    """a(x) * x^e, e taken mod 2N (x^-k = x^(2N-k))."""
    e %= 2 * N
    out = [0] * N
    for i, c in enumerate(a):
        d = (i + e) % (2 * N)
        out[d % N] = (-int(c)) % q if d >= N else int(c) % q
    return out


def crt(residues: Sequence[int], moduli: Sequence[int]) -> int:
    M = 1
    for m in moduli:
        M *= m
    x = 0
    for r, m in zip(residues, moduli):
        Mi = M // m
        x += int(r) * Mi * pow(Mi, -1, m)
    return x % M


class Model:
    """moduli = k data primes + the special prime; ct = [2][k][N] lists; key = [k][2][k+1][N] (NTT form)."""

    def __init__(self, N: int, moduli: Sequence[int]):
        self.N = N
        self.moduli = [int(m) for m in moduli]
        self.k = len(moduli) - 1
        self.rings = [Ring(N, q) for q in self.moduli]

    # --- App. A.4 --------------------------------------------------------------------------------
    def key_switch_acc(self, target, key, comp):
        """S_i = sum_j (D_j mod m_i) * INTT_i(K[j][comp][i]) over every key-level modulus i (coefficient form)."""
        k, N = self.k, self.N
        S = []
        for i, ring in enumerate(self.rings):
            acc = [0] * N
            for j in range(k):
                dj = [int(v) % ring.q for v in target[j]]
                kj = ring.intt(key[j][comp][i])
                prod = ring.mul(dj, kj)
                acc = [(a + b) % ring.q for a, b in zip(acc, prod)]
            S.append(acc)
        return S

    def mod_down_rns(self, S):
        """App. A.4 formula, modulus by modulus."""
        k, p = self.k, self.moduli[-1]
        half = p // 2
        out = []
        for i in range(k):
            qi = self.moduli[i]
            pinv = pow(p, -1, qi)
            row = []
            for c in range(self.N):
                r = (S[k][c] + half) % p
                delta = (r % qi) - (half % qi)
                row.append(((S[i][c] - delta) * pinv) % qi)
            out.append(row)
        return out

    def mod_down_bigint(self, S):
        """floor((X + floor(p/2)) / p) mod q_i with X the CRT lift to [0, Q*p)."""
        k, p = self.k, self.moduli[-1]
        half = p // 2
        out = [[0] * self.N for _ in range(k)]
        for c in range(self.N):
            X = crt([S[i][c] for i in range(k + 1)], self.moduli)
            y = (X + half) // p
            for i in range(k):
                out[i][c] = y % self.moduli[i]
        return out

    def switch_key(self, target, key, bigint=False):
        """-> [2][k][N]: the two polynomials key switching adds to the ciphertext."""
        out = []
        for comp in range(2):
            S = self.key_switch_acc(target, key, comp)
            out.append(self.mod_down_bigint(S) if bigint else self.mod_down_rns(S))
        return out

    # --- App. A.3: Evaluator::apply_galois_inplace -------------------------------------------------
    def apply_galois_ct(self, ct, g, key, bigint=False):
        k, N = self.k, self.N
        c0 = [apply_galois(ct[0][j], g, N, self.moduli[j]) for j in range(k)]
        c1 = [apply_galois(ct[1][j], g, N, self.moduli[j]) for j in range(k)]
        ks = self.switch_key(c1, key, bigint)
        new0 = [[(a + b) % self.moduli[j] for a, b in zip(c0[j], ks[0][j])] for j in range(k)]
        return [new0, ks[1]]

    def ct_add(self, a, b):
        return [[[(x + y) % self.moduli[j] for x, y in zip(a[p][j], b[p][j])] for j in range(self.k)] for p in range(2)]

    def ct_mul_monomial(self, ct, e):
        return [[mul_monomial(ct[p][j], e, self.N, self.moduli[j]) for j in range(self.k)] for p in range(2)]

    # --- reference server.cpp:105-146 ----------------------------------------------------------------
    def oblivious_expansion(self, ct, n: int, keys: Dict[int, list], bigint=False):
        N = self.N
        logm = max(n - 1, 0).bit_length()
        m = 1 << logm
        results = [None] * m
        results[0] = ct
        for j in range(logm):
            two_j = 1 << j
            for kk in range(two_j):
                c0 = results[kk]
                g = (N >> j) + 1
                c0s = self.apply_galois_ct(c0, g, keys[g], bigint)          # :123-126
                results[kk + two_j] = self.ct_mul_monomial(results[kk], -two_j)          # :129-130
                c1 = self.ct_mul_monomial(c0s, -(N + two_j))                # :137-138
                results[kk] = self.ct_add(results[kk], c0s)                 # :140
                results[kk + two_j] = self.ct_add(results[kk + two_j], c1)  # :141
        return results[:n]

    # --- App. A.5 + reference database.cpp:170-258, ct_reencoder.cpp:29-71 ------------------------
    def plain_lift(self, coeffs, t: int):
        """Evaluator::transform_to_ntt_inplace(Plaintext)'s lift, kept in coefficient form: m >= (t+1)/2 -> m + q_j - t."""
        thr = (t + 1) >> 1
        out = []
        for j in range(self.k):
            q = self.moduli[j]
            row = [((int(m) + q - t) % q if int(m) >= thr else int(m) % q) for m in coeffs]
            out.append(row + [0] * (self.N - len(row)))
        return out

    def multiply_plain(self, ct, pt_lifted):
        """ct (coefficient form) times a lifted plaintext: the ring product the NTT-domain dyadic product computes."""
        return [[self.rings[j].mul(ct[p][j], pt_lifted[j]) for j in range(self.k)] for p in range(2)]

    def expansion_ratio(self, t: int) -> int:
        import math
        b = int(math.log2(t))
        return sum(int(math.ceil(math.log2(self.moduli[j]) / b)) for j in range(self.k))

    def reencode(self, ct, t: int):
        import math
        b = int(math.log2(t))
        mask = (1 << b) - 1
        out = []
        for p in range(2):
            for j in range(self.k):
                ler = int(math.ceil(math.log2(self.moduli[j]) / b))
                for i in range(ler):
                    out.append([(int(v) >> (i * b)) & mask for v in ct[p][j]])
        return out

    def db_multiply(self, db_coeffs, dims, sv, t: int):
        """PIRDatabase::multiply (decomposition mode): db_coeffs = plaintext coefficient lists (< t), sv =
        dim_sum ciphertexts in coefficient form.  Returns the reply ciphertexts, coefficient form."""
        pos = [0]

        def rec(dims_, sv_off):
            this = dims_[0]
            rest = dims_[1:]
            result = None
            for i in range(this):
                if pos[0] >= len(db_coeffs):
                    break
                if not rest:
                    temp = [self.multiply_plain(sv[sv_off + i], self.plain_lift(db_coeffs[pos[0]], t))]
                    pos[0] += 1
                else:
                    lower = rec(rest, sv_off + this)
                    temp = []
                    for ct in lower:
                        for pt in self.reencode(ct, t):
                            temp.append(self.multiply_plain(sv[sv_off + i], self.plain_lift(pt, t)))
                result = temp if result is None else [self.ct_add(a, b) for a, b in zip(result, temp)]
            return result

        return rec(list(dims), 0)
