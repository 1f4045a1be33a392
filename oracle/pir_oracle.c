/*
 * pir_oracle.c -- CPU restatement of the OpenMined/PIR server query path.
 *
 * TEST INFRASTRUCTURE ONLY (see pir_oracle.h).  "parity unpinned" at the
 * ciphertext-bit level: SEAL 3.5.6 is not available here; the restatement is
 * pinned on the reference tests' plaintext-level known answers instead.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference) or the SEAL 3.5.6 primitive whose published semantics it
 * follows.  All public results are canonical residues in
 * [0, q), exactly as SEAL's public entry points return them, so any correct
 * implementation of the same formulas produces the same bits.
 */
#include "pir_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

struct orc_ctx {
  uint32_t N, logN, k;
  int has_special;
  uint64_t q[ORC_MAXK + 1];
  uint64_t t;
  uint64_t psi[ORC_MAXK + 1];
  /* psi^bitrev(i) and psi^-bitrev(i) with Shoup quotients floor(w * 2^64 / q) */
  uint64_t* w[ORC_MAXK + 1];
  uint64_t* ws[ORC_MAXK + 1];
  uint64_t* iw[ORC_MAXK + 1];
  uint64_t* iws[ORC_MAXK + 1];
  uint64_t ninv[ORC_MAXK + 1], ninvs[ORC_MAXK + 1];
  /* key-switch mod-down constants */
  uint64_t pinv[ORC_MAXK];   /* p^-1 mod q_j */
  uint64_t phalf_mod[ORC_MAXK]; /* floor(p/2) mod q_j */
  /* plain lift: q_j - (t mod q_j) */
  uint64_t lift_inc[ORC_MAXK];
  /* Barrett ratio floor(2^128 / q) per modulus (SEAL Modulus::const_ratio), for the dyadic products */
  uint64_t br_lo[ORC_MAXK + 1], br_hi[ORC_MAXK + 1];
};

/* ---------------------------------------------------------------- arithmetic */

uint64_t orc_mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }

uint64_t orc_powmod(uint64_t a, uint64_t e, uint64_t q) {
  uint64_t r = 1 % q;
  a %= q;
  while (e) {
    if (e & 1) r = orc_mulmod(r, a, q);
    a = orc_mulmod(a, a, q);
    e >>= 1;
  }
  return r;
}

uint64_t orc_invmod(uint64_t a, uint64_t q) { return orc_powmod(a, q - 2, q); }

int orc_is_prime(uint64_t n) {
  if (n < 2) return 0;
  static const uint64_t small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  for (size_t i = 0; i < sizeof(small) / sizeof(small[0]); ++i) {
    if (n == small[i]) return 1;
    if (n % small[i] == 0) return 0;
  }
  uint64_t d = n - 1;
  int r = 0;
  while ((d & 1) == 0) {
    d >>= 1;
    ++r;
  }
  /* deterministic Miller-Rabin for 64-bit n */
  for (size_t i = 0; i < sizeof(small) / sizeof(small[0]); ++i) {
    uint64_t x = orc_powmod(small[i], d, n);
    if (x == 1 || x == n - 1) continue;
    int comp = 1;
    for (int j = 1; j < r; ++j) {
      x = orc_mulmod(x, x, n);
      if (x == n - 1) {
        comp = 0;
        break;
      }
    }
    if (comp) return 0;
  }
  return 1;
}

/* SEAL util::try_minimal_primitive_root: the smallest primitive two_n-th root
 * of unity mod q.  Deterministic: the minimum over all odd
 * powers of any primitive root. */
uint64_t orc_minimal_primitive_root(uint64_t two_n, uint64_t q) {
  if ((q - 1) % two_n) return 0;
  uint64_t e = (q - 1) / two_n, root = 0;
  for (uint64_t x = 2; x < q; ++x) {
    uint64_t r = orc_powmod(x, e, q);
    if (orc_powmod(r, two_n / 2, q) == q - 1) {
      root = r;
      break;
    }
  }
  if (!root) return 0;
  uint64_t sq = orc_mulmod(root, root, q), cur = root, best = root;
  for (uint64_t i = 0; i < two_n / 2; ++i) {
    if (cur < best) best = cur;
    cur = orc_mulmod(cur, sq, q);
  }
  return best;
}

static inline uint64_t shoup(uint64_t w, uint64_t q) { return (uint64_t)(((u128)w << 64) / q); }

/* x * w mod q with precomputed ws = floor(w 2^64 / q); x arbitrary 64-bit. */
static inline uint64_t mul_shoup(uint64_t x, uint64_t w, uint64_t ws, uint64_t q) {
  uint64_t h = (uint64_t)(((u128)x * ws) >> 64);
  uint64_t r = x * w - h * q;
  return r >= q ? r - q : r;
}

static inline uint64_t addmod(uint64_t a, uint64_t b, uint64_t q) {
  uint64_t s = a + b;
  return s >= q ? s - q : s;
}
static inline uint64_t submod(uint64_t a, uint64_t b, uint64_t q) { return a >= b ? a - b : a + q - b; }

static uint32_t bitrev(uint32_t x, uint32_t bits) {
  uint32_t r = 0;
  for (uint32_t i = 0; i < bits; ++i) {
    r = (r << 1) | (x & 1);
    x >>= 1;
  }
  return r;
}

/* ------------------------------------------------------------------ context */

orc_ctx* orc_create(uint32_t N, uint32_t k, const uint64_t* moduli, uint64_t t) {
  if (k == 0 || k > ORC_MAXK || N < 2 || (N & (N - 1))) return NULL;
  orc_ctx* c = (orc_ctx*)calloc(1, sizeof(orc_ctx));
  c->N = N;
  c->k = k;
  c->t = t;
  while ((1u << c->logN) < N) ++c->logN;
  c->has_special = moduli[k] != 0;
  uint32_t nm = k + (c->has_special ? 1 : 0);
  for (uint32_t i = 0; i < nm; ++i) {
    uint64_t q = moduli[i];
    c->q[i] = q;
    if (q >> 61 || !orc_is_prime(q) || (q - 1) % (2ull * N)) {
      orc_destroy(c);
      return NULL;
    }
    /* SEAL NTTTables::initialize: root powers stored in bit-reversed order */
    uint64_t psi = orc_minimal_primitive_root(2ull * N, q);
    uint64_t ipsi = orc_invmod(psi, q);
    c->psi[i] = psi;
    c->w[i] = (uint64_t*)malloc(sizeof(uint64_t) * N);
    c->ws[i] = (uint64_t*)malloc(sizeof(uint64_t) * N);
    c->iw[i] = (uint64_t*)malloc(sizeof(uint64_t) * N);
    c->iws[i] = (uint64_t*)malloc(sizeof(uint64_t) * N);
    uint64_t p = 1, ip = 1;
    for (uint32_t j = 0; j < N; ++j) {
      uint32_t r = bitrev(j, c->logN);
      c->w[i][r] = p;
      c->ws[i][r] = shoup(p, q);
      c->iw[i][r] = ip;
      c->iws[i][r] = shoup(ip, q);
      p = orc_mulmod(p, psi, q);
      ip = orc_mulmod(ip, ipsi, q);
    }
    c->ninv[i] = orc_invmod(N % q, q);
    {
      const u128 ratio = (~(u128)0) / q; /* floor((2^128 - 1) / q) == floor(2^128 / q) for odd q > 1 */
      c->br_lo[i] = (uint64_t)ratio;
      c->br_hi[i] = (uint64_t)(ratio >> 64);
    }
    c->ninvs[i] = shoup(c->ninv[i], q);
  }
  for (uint32_t j = 0; j < k; ++j) {
    c->lift_inc[j] = c->q[j] - (t % c->q[j]);
    if (c->has_special) {
      uint64_t p = c->q[k];
      c->pinv[j] = orc_invmod(p % c->q[j], c->q[j]);
      c->phalf_mod[j] = (p >> 1) % c->q[j];
    }
  }
  return c;
}

void orc_destroy(orc_ctx* c) {
  if (!c) return;
  for (uint32_t i = 0; i <= ORC_MAXK; ++i) {
    free(c->w[i]);
    free(c->ws[i]);
    free(c->iw[i]);
    free(c->iws[i]);
  }
  free(c);
}

uint32_t orc_N(const orc_ctx* c) { return c->N; }
uint32_t orc_k(const orc_ctx* c) { return c->k; }
uint64_t orc_modulus(const orc_ctx* c, uint32_t i) { return c->q[i]; }
uint64_t orc_plain_modulus(const orc_ctx* c) { return c->t; }
uint64_t orc_psi(const orc_ctx* c, uint32_t i) { return c->psi[i]; }

/* ---------------------------------------------------------------------- NTT */

/* SEAL ntt_negacyclic_harvey: Cooley-Tukey, natural-order
 * input, bit-reversed-order output, twiddle table psi^bitrev(m+i). */
void orc_ntt_fwd(const orc_ctx* c, uint32_t mi, uint64_t* a) {
  const uint64_t q = c->q[mi];
  const uint64_t *w = c->w[mi], *ws = c->ws[mi];
  uint32_t n = c->N, t = n;
  for (uint32_t m = 1; m < n; m <<= 1) {
    t >>= 1;
    for (uint32_t i = 0; i < m; ++i) {
      uint64_t W = w[m + i], Ws = ws[m + i];
      uint64_t* x = a + 2 * i * t;
      uint64_t* y = x + t;
      for (uint32_t j = 0; j < t; ++j) {
        uint64_t u = x[j], v = mul_shoup(y[j], W, Ws, q);
        x[j] = addmod(u, v, q);
        y[j] = submod(u, v, q);
      }
    }
  }
}

/* SEAL inverse_ntt_negacyclic_harvey: Gentleman-Sande, bit-reversed input,
 * natural output, scaled by N^-1. */
void orc_ntt_inv(const orc_ctx* c, uint32_t mi, uint64_t* a) {
  const uint64_t q = c->q[mi];
  const uint64_t *iw = c->iw[mi], *iws = c->iws[mi];
  uint32_t n = c->N, t = 1;
  for (uint32_t m = n; m > 1; m >>= 1) {
    uint32_t h = m >> 1;
    for (uint32_t i = 0; i < h; ++i) {
      uint64_t W = iw[h + i], Ws = iws[h + i];
      uint64_t* x = a + 2 * i * t;
      uint64_t* y = x + t;
      for (uint32_t j = 0; j < t; ++j) {
        uint64_t u = x[j], v = y[j];
        x[j] = addmod(u, v, q);
        y[j] = mul_shoup(submod(u, v, q), W, Ws, q);
      }
    }
    t <<= 1;
  }
  for (uint32_t j = 0; j < n; ++j) a[j] = mul_shoup(a[j], c->ninv[mi], c->ninvs[mi], q);
}

/* Evaluator::transform_to_ntt_inplace(Ciphertext&) -- reference database.cpp:190,222 */
void orc_ct_ntt_fwd(const orc_ctx* c, uint64_t* ct) {
  for (uint32_t p = 0; p < 2; ++p)
    for (uint32_t j = 0; j < c->k; ++j) orc_ntt_fwd(c, j, ct + ((size_t)p * c->k + j) * c->N);
}
/* Evaluator::transform_from_ntt_inplace -- reference database.cpp:252 */
void orc_ct_ntt_inv(const orc_ctx* c, uint64_t* ct) {
  for (uint32_t p = 0; p < 2; ++p)
    for (uint32_t j = 0; j < c->k; ++j) orc_ntt_inv(c, j, ct + ((size_t)p * c->k + j) * c->N);
}

/* (hi:lo) mod q with the precomputed ratio floor(2^128 / q): SEAL 3.5.6 util::barrett_reduce_128 -- what
 * Evaluator::multiply_plain's dyadic product (util::dyadic_product_coeffmod) uses; canonical result. */
static inline uint64_t barrett_reduce_128(uint64_t lo, uint64_t hi, uint64_t q, uint64_t br_lo, uint64_t br_hi) {
  const uint64_t carry = (uint64_t)(((u128)lo * br_lo) >> 64);
  const u128 t2 = (u128)lo * br_hi;
  const uint64_t t1 = (uint64_t)t2 + carry;
  const uint64_t t3 = (uint64_t)(t2 >> 64) + (t1 < (uint64_t)t2);
  const u128 t4 = (u128)hi * br_lo;
  const uint64_t t1b = t1 + (uint64_t)t4;
  const uint64_t carry2 = (uint64_t)(t4 >> 64) + (t1b < t1);
  const uint64_t qhat = hi * br_hi + t3 + carry2;
  const uint64_t r = lo - qhat * q;
  return r >= q ? r - q : r;
}

void orc_dyadic_mul(const orc_ctx* c, uint32_t mi, const uint64_t* a, const uint64_t* b, uint64_t* out) {
  const uint64_t q = c->q[mi], bl = c->br_lo[mi], bh = c->br_hi[mi];
  for (uint32_t i = 0; i < c->N; ++i) {
    const u128 z = (u128)a[i] * b[i];
    out[i] = barrett_reduce_128((uint64_t)z, (uint64_t)(z >> 64), q, bl, bh);
  }
}
void orc_poly_add(const orc_ctx* c, uint32_t mi, const uint64_t* a, const uint64_t* b, uint64_t* out) {
  for (uint32_t i = 0; i < c->N; ++i) out[i] = addmod(a[i], b[i], c->q[mi]);
}
void orc_poly_sub(const orc_ctx* c, uint32_t mi, const uint64_t* a, const uint64_t* b, uint64_t* out) {
  for (uint32_t i = 0; i < c->N; ++i) out[i] = submod(a[i], b[i], c->q[mi]);
}
void orc_poly_neg(const orc_ctx* c, uint32_t mi, const uint64_t* a, uint64_t* out) {
  for (uint32_t i = 0; i < c->N; ++i) out[i] = a[i] ? c->q[mi] - a[i] : 0;
}

/* ------------------------------------------------- Galois / monomial / add */

/* SEAL GaloisTool::apply_galois, coefficient form:
 * out[(i*g) mod N] = +-in[i], negated when floor(i*g / N) is odd. */
void orc_apply_galois_poly(const orc_ctx* c, uint32_t mi, const uint64_t* in, uint32_t g, uint64_t* out) {
  const uint64_t q = c->q[mi];
  const uint32_t N = c->N, logN = c->logN;
  for (uint32_t i = 0; i < N; ++i) {
    uint64_t raw = (uint64_t)i * g;
    uint32_t idx = (uint32_t)(raw & (N - 1));
    uint64_t v = in[i];
    if ((raw >> logN) & 1) v = v ? q - v : 0;
    out[idx] = v;
  }
}

/* SEAL util::negacyclic_shift_poly_coeffmod. */
void orc_negacyclic_shift_poly(const orc_ctx* c, uint32_t mi, const uint64_t* in, uint32_t shift, uint64_t* out) {
  const uint64_t q = c->q[mi];
  const uint32_t N = c->N;
  if (shift == 0) {
    memcpy(out, in, sizeof(uint64_t) * N);
    return;
  }
  for (uint32_t i = 0; i < N; ++i) {
    uint64_t raw = (uint64_t)i + shift;
    uint32_t idx = (uint32_t)(raw & (N - 1));
    uint64_t v = in[i];
    if ((raw & N) && v) v = q - v;
    out[idx] = v;
  }
}

/* SEAL RNSTool::divide_and_round_q_last: round(x / p) per data residue
 * (second half). */
void orc_divide_round_special(const orc_ctx* c, const uint64_t* in, uint64_t* out) {
  const uint32_t N = c->N, k = c->k;
  const uint64_t p = c->q[k], half = p >> 1;
  const uint64_t* last = in + (size_t)k * N;
  for (uint32_t j = 0; j < k; ++j) {
    const uint64_t q = c->q[j];
    for (uint32_t i = 0; i < N; ++i) {
      uint64_t r = addmod(last[i], half % p, p); /* (x_p + floor(p/2)) mod p */
      uint64_t delta = submod(r % q, c->phalf_mod[j], q);
      uint64_t v = submod(in[(size_t)j * N + i], delta, q);
      out[(size_t)j * N + i] = orc_mulmod(v, c->pinv[j], q);
    }
  }
}

/* Evaluator::switch_key_inplace for BFV: target = [k][N]
 * coefficient form; key = [k][2][k+1][N] NTT form; adds the switched pair to ct. */
static void switch_key_inplace(const orc_ctx* c, uint64_t* ct, const uint64_t* target, const uint64_t* key) {
  const uint32_t N = c->N, k = c->k, km = k + 1;
  uint64_t* tmp = (uint64_t*)malloc(sizeof(uint64_t) * N);
  u128* acc = (u128*)malloc(sizeof(u128) * 2 * N);
  uint64_t* prod = (uint64_t*)malloc(sizeof(uint64_t) * 2 * km * N); /* [comp][km][N] */
  for (uint32_t I = 0; I < km; ++I) {
    const uint64_t m = c->q[I];
    memset(acc, 0, sizeof(u128) * 2 * N);
    for (uint32_t J = 0; J < k; ++J) {
      const uint64_t* src = target + (size_t)J * N;
      for (uint32_t i = 0; i < N; ++i) tmp[i] = src[i] % m;
      orc_ntt_fwd(c, I, tmp);
      for (uint32_t comp = 0; comp < 2; ++comp) {
        const uint64_t* kk = key + (((size_t)J * 2 + comp) * km + I) * N;
        u128* a = acc + (size_t)comp * N;
        for (uint32_t i = 0; i < N; ++i) a[i] += (u128)tmp[i] * kk[i];
      }
    }
    for (uint32_t comp = 0; comp < 2; ++comp) {
      uint64_t* dst = prod + ((size_t)comp * km + I) * N;
      for (uint32_t i = 0; i < N; ++i) {   /* lazy 128-bit sums, one Barrett reduction (as SEAL's key switch) */
        const u128 a = acc[(size_t)comp * N + i];
        dst[i] = barrett_reduce_128((uint64_t)a, (uint64_t)(a >> 64), m, c->br_lo[I], c->br_hi[I]);
      }
      orc_ntt_inv(c, I, dst);
    }
  }
  uint64_t* down = (uint64_t*)malloc(sizeof(uint64_t) * k * N);
  for (uint32_t comp = 0; comp < 2; ++comp) {
    orc_divide_round_special(c, prod + (size_t)comp * km * N, down);
    for (uint32_t j = 0; j < k; ++j) {
      uint64_t* dst = ct + ((size_t)comp * k + j) * N;
      orc_poly_add(c, j, dst, down + (size_t)j * N, dst);
    }
  }
  free(down);
  free(prod);
  free(acc);
  free(tmp);
}

/* reference server.cpp:67-76 -> Evaluator::apply_galois_inplace:
 * ct <- (sigma_g(c0), 0) + KeySwitch(sigma_g(c1)). */
int orc_apply_galois_ct(const orc_ctx* c, uint64_t* ct, uint32_t g, const uint64_t* key) {
  const uint32_t N = c->N, k = c->k;
  if (!c->has_special || !key || !(g & 1) || g >= 2 * N) return ORC_INTERNAL;
  uint64_t* tmp = (uint64_t*)malloc(sizeof(uint64_t) * k * N);
  for (uint32_t j = 0; j < k; ++j) orc_apply_galois_poly(c, j, ct + (size_t)j * N, g, tmp + (size_t)j * N);
  memcpy(ct, tmp, sizeof(uint64_t) * k * N);
  for (uint32_t j = 0; j < k; ++j)
    orc_apply_galois_poly(c, j, ct + ((size_t)k + j) * N, g, tmp + (size_t)j * N);
  memset(ct + (size_t)k * N, 0, sizeof(uint64_t) * k * N);
  switch_key_inplace(c, ct, tmp, key);
  free(tmp);
  return ORC_OK;
}

/* reference server.cpp:78-103 */
void orc_multiply_inverse_power_of_x(const orc_ctx* c, const uint64_t* ct, uint32_t kpow, uint64_t* out) {
  const uint32_t N = c->N, k = c->k;
  uint32_t index = ((N << 1) - kpow) % (N << 1);
  for (uint32_t p = 0; p < 2; ++p)
    for (uint32_t j = 0; j < k; ++j)
      orc_negacyclic_shift_poly(c, j, ct + ((size_t)p * k + j) * N, index, out + ((size_t)p * k + j) * N);
}

void orc_ct_add_inplace(const orc_ctx* c, uint64_t* a, const uint64_t* b) {
  for (uint32_t p = 0; p < 2; ++p)
    for (uint32_t j = 0; j < c->k; ++j) {
      size_t o = ((size_t)p * c->k + j) * c->N;
      orc_poly_add(c, j, a + o, b + o, a + o);
    }
}

/* --------------------------------------------------------------- utilities */

/* reference utils.cpp:16-44 (values only; the De Bruijn tables are an
 * implementation detail of the same functions). */
uint32_t orc_log2(uint32_t v) {
  uint32_t r = 0;
  while (v >>= 1) ++r;
  return r;
}
uint32_t orc_ceil_log2(uint32_t v) {
  if (v <= 1) return 0;
  return orc_log2(v - 1) + 1;
}
/* reference utils.h:29-37 */
uint64_t orc_next_power_two(uint64_t n) {
  if (n == 0) return 1;
  uint64_t p = 1;
  while (p < n) p <<= 1;
  return p;
}

/* ------------------------------------------------------ oblivious expansion */

/* reference server.cpp:105-146 */
int orc_oblivious_expansion(const orc_ctx* c, const uint64_t* ct, uint32_t num_items,
                            const uint64_t* const* galois_keys, uint64_t* out) {
  const uint32_t N = c->N, k = c->k;
  const size_t ctw = (size_t)2 * k * N;
  if (num_items > N) return ORC_INVALID_ARGUMENT;
  uint32_t logm = orc_ceil_log2(num_items);
  uint32_t m = (uint32_t)orc_next_power_two(num_items);
  uint64_t* res = (uint64_t*)malloc(sizeof(uint64_t) * ctw * m);
  uint64_t* c0 = (uint64_t*)malloc(sizeof(uint64_t) * ctw);
  uint64_t* c1 = (uint64_t*)malloc(sizeof(uint64_t) * ctw);
  memcpy(res, ct, sizeof(uint64_t) * ctw);
  int rc = ORC_OK;
  for (uint32_t j = 0; j < logm && rc == ORC_OK; ++j) {
    const uint32_t two_j = 1u << j;
    for (uint32_t kk = 0; kk < two_j; ++kk) {
      uint64_t* rk = res + ctw * kk;
      uint64_t* rk2 = res + ctw * (kk + two_j);
      memcpy(c0, rk, sizeof(uint64_t) * ctw);
      rc = orc_apply_galois_ct(c, c0, (N >> j) + 1, galois_keys ? galois_keys[j] : NULL);
      if (rc != ORC_OK) break;
      orc_multiply_inverse_power_of_x(c, rk, two_j, rk2);
      orc_multiply_inverse_power_of_x(c, c0, N + two_j, c1);
      orc_ct_add_inplace(c, rk, c0);
      orc_ct_add_inplace(c, rk2, c1);
    }
  }
  if (rc == ORC_OK) memcpy(out, res, sizeof(uint64_t) * ctw * num_items);
  free(c1);
  free(c0);
  free(res);
  return rc;
}

/* reference server.cpp:148-171 */
int orc_oblivious_expansion_multi(const orc_ctx* c, const uint64_t* cts, uint32_t num_cts, uint64_t total_items,
                                  const uint64_t* const* galois_keys, uint64_t* out) {
  const uint32_t N = c->N;
  const size_t ctw = (size_t)2 * c->k * N;
  if (num_cts != total_items / N + 1) return ORC_INVALID_ARGUMENT;
  uint64_t remaining = total_items;
  size_t produced = 0;
  for (uint32_t i = 0; i < num_cts; ++i) {
    /* size_t arithmetic as in the reference: after the last full ciphertext the
     * remaining count may be 0 (total % N == 0) -> zero-item expansion. */
    uint32_t n = (uint32_t)(remaining < N ? remaining : N);
    if (n > 0) {
      int rc = orc_oblivious_expansion(c, cts + ctw * i, n, galois_keys, out + ctw * produced);
      if (rc != ORC_OK) return rc;
    }
    produced += n;
    remaining -= N; /* wraps like the reference's size_t; unused afterwards */
    if (produced >= total_items) break;
  }
  return ORC_OK;
}

/* --------------------------------------------------------- plaintext ops */

/* Evaluator::transform_to_ntt_inplace(Plaintext&, parms_id):
 * m >= (t+1)/2 ? m + (q_j - t) : m, then forward NTT per residue. */
void orc_plain_lift_ntt(const orc_ctx* c, const uint64_t* coeffs, uint32_t ncoeff, uint64_t* out) {
  const uint32_t N = c->N;
  const uint64_t thr = (c->t + 1) >> 1;
  for (uint32_t j = 0; j < c->k; ++j) {
    uint64_t* dst = out + (size_t)j * N;
    const uint64_t q = c->q[j];
    for (uint32_t i = 0; i < N; ++i) {
      uint64_t m = i < ncoeff ? coeffs[i] : 0;
      dst[i] = m >= thr ? addmod(m % q, c->lift_inc[j] % q, q) : m % q;
    }
    orc_ntt_fwd(c, j, dst);
  }
}

/* Evaluator::multiply_plain, NTT x NTT operands (reference database.cpp:192,229). */
void orc_multiply_plain_ntt(const orc_ctx* c, const uint64_t* ct, const uint64_t* pt, uint64_t* out) {
  for (uint32_t p = 0; p < 2; ++p)
    for (uint32_t j = 0; j < c->k; ++j)
      orc_dyadic_mul(c, j, ct + ((size_t)p * c->k + j) * c->N, pt + (size_t)j * c->N,
                     out + ((size_t)p * c->k + j) * c->N);
}

/* ------------------------------------------------------------- re-encoder */

/* reference ct_reencoder.cpp:32,42,80 / string_encoder.cpp:85: <cmath> log2 of
 * the plain modulus truncated to an integer. */
uint32_t orc_bits_per_coeff(uint64_t t) { return (uint32_t)log2((double)t); }

static uint32_t local_expansion_ratio(uint64_t q, uint32_t b) {
  double bits = log2((double)q);
  return (uint32_t)ceil(bits / b);
}

/* reference ct_reencoder.cpp:29-38 */
uint32_t orc_expansion_ratio(const orc_ctx* c) {
  uint32_t b = orc_bits_per_coeff(c->t), er = 0;
  for (uint32_t j = 0; j < c->k; ++j) er += local_expansion_ratio(c->q[j], b);
  return er;
}

/* reference ct_reencoder.cpp:40-71 (ct.size() == 2).  The mask `(1 << b) - 1` is `int` arithmetic there (:45):
 * identical to the 64-bit expression for b < 31, UNDEFINED BEHAVIOUR for b >= 31 (a shift by the width of int or more;
 * gcc on x86-64 happens to emit a shift by b mod 32, i.e. a mask of 9 bits at b = 41).  The reference's own tests
 * reach such b only with CT multiplication switched on (correctness_test.cpp:99: 42-bit t), where the re-encoder is
 * not used.  For b >= 31 the restatement therefore takes the mathematically intended mask 2^b - 1 -- the only choice
 * under which Decode (:73-112) inverts Encode -- instead of reproducing one compiler's treatment of the UB. */
void orc_reencode(const orc_ctx* c, const uint64_t* ct, uint64_t* pts) {
  const uint32_t N = c->N, k = c->k, b = orc_bits_per_coeff(c->t);
  const uint64_t mask = b >= 31 ? ((uint64_t)1 << b) - 1 : (uint64_t)((1 << b) - 1);
  uint64_t* dst = pts;
  for (uint32_t p = 0; p < 2; ++p)
    for (uint32_t j = 0; j < k; ++j) {
      uint32_t ler = local_expansion_ratio(c->q[j], b);
      uint32_t shift = 0;
      for (uint32_t e = 0; e < ler; ++e) {
        const uint64_t* src = ct + ((size_t)p * k + j) * N;
        for (uint32_t i = 0; i < N; ++i) dst[i] = (src[i] >> shift) & mask;
        dst += N;
        shift += b;
      }
    }
}

/* reference ct_reencoder.cpp:73-112 (client side; used by the test client) */
void orc_redecode(const orc_ctx* c, const uint64_t* pts, const uint32_t* pt_ncoeff, uint64_t* ct) {
  const uint32_t N = c->N, k = c->k, b = orc_bits_per_coeff(c->t);
  const uint64_t* src = pts;
  uint32_t idx = 0;
  memset(ct, 0, sizeof(uint64_t) * 2 * k * N);
  for (uint32_t p = 0; p < 2; ++p)
    for (uint32_t j = 0; j < k; ++j) {
      uint32_t ler = local_expansion_ratio(c->q[j], b);
      uint32_t shift = 0;
      uint64_t* dst = ct + ((size_t)p * k + j) * N;
      for (uint32_t e = 0; e < ler; ++e) {
        uint32_t nc = pt_ncoeff ? pt_ncoeff[idx] : N;
        for (uint32_t i = 0; i < nc; ++i) {
          if (shift == 0)
            dst[i] = src[i];
          else
            dst[i] += src[i] << shift;
        }
        src += N;
        ++idx;
        shift += b;
      }
    }
}

/* ---------------------------------------------------- database multiply */

uint64_t orc_reply_ct_count(const orc_ctx* c, uint32_t nd) {
  uint64_t r = 1, f = (uint64_t)2 * orc_expansion_ratio(c);
  for (uint32_t i = 1; i < nd; ++i) r *= f;
  return r;
}

typedef struct {
  const orc_ctx* c;
  const uint64_t* db;
  uint64_t P, pos; /* database_it_ */
  uint64_t* sv;
  uint8_t* sv_ntt;
  uint32_t er;
  int transparent; /* a multiply_plain / add_inplace result had an all-zero c1: SEAL throws (App. A.5) */
} dbm_t;

/* Ciphertext::is_transparent for a size-2 ciphertext: every coefficient of c1 is zero.  SEAL's default build
 * (SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT) makes Evaluator::multiply_plain / add_inplace throw
 * logic_error("result ciphertext is transparent") on such a result; reference database.cpp:308-315 turns any
 * exception of the multiplication into InternalError. */
static int ct_is_transparent(const orc_ctx* c, const uint64_t* ct) {
  const size_t n = (size_t)c->k * c->N;
  const uint64_t* c1 = ct + n;
  for (size_t i = 0; i < n; ++i)
    if (c1[i]) return 0;
  return 1;
}

static void ensure_sv_ntt(dbm_t* m, uint64_t idx) {
  if (!m->sv_ntt[idx]) {
    orc_ct_ntt_fwd(m->c, m->sv + idx * 2 * m->c->k * m->c->N);
    m->sv_ntt[idx] = 1;
  }
}

/* reference database.cpp:170-258 (DatabaseMultiplier::multiply, decomposition
 * mode).  Returns a malloc'd array of *count ciphertexts in coefficient form. */
static uint64_t* dbm_multiply(dbm_t* m, const uint32_t* dims, uint32_t nd, uint64_t sv_off, uint64_t* count) {
  const orc_ctx* c = m->c;
  const uint32_t N = c->N, k = c->k;
  const size_t ctw = (size_t)2 * k * N;
  const uint32_t this_dim = dims[0];
  uint64_t* result = NULL;
  uint64_t rcount = 0;
  uint64_t* temp = NULL;
  int first = 1;
  for (uint32_t i = 0; i < this_dim; ++i) {
    if (m->pos == m->P) break; /* database.cpp:183 */
    uint64_t tcount;
    if (nd == 1) {
      /* base case, database.cpp:185-194 */
      tcount = 1;
      if (!temp) temp = (uint64_t*)malloc(sizeof(uint64_t) * ctw);
      ensure_sv_ntt(m, sv_off + i);
      orc_multiply_plain_ntt(c, m->sv + (sv_off + i) * ctw, m->db + m->pos * (size_t)k * N, temp);
      if (ct_is_transparent(c, temp)) m->transparent = 1;
      ++m->pos;
    } else {
      uint64_t lcount;
      uint64_t* lower = dbm_multiply(m, dims + 1, nd - 1, sv_off + this_dim, &lcount);
      tcount = lcount * m->er * 2; /* database.cpp:214 */
      if (!temp) temp = (uint64_t*)malloc(sizeof(uint64_t) * ctw * tcount);
      uint64_t* pts = (uint64_t*)malloc(sizeof(uint64_t) * 2 * m->er * N);
      uint64_t* ptn = (uint64_t*)malloc(sizeof(uint64_t) * k * N);
      uint64_t ti = 0;
      for (uint64_t l = 0; l < lcount; ++l) {
        orc_reencode(c, lower + l * ctw, pts); /* database.cpp:218 */
        for (uint32_t e = 0; e < 2 * m->er; ++e) {
          ensure_sv_ntt(m, sv_off + i);                      /* :221-224 */
          orc_plain_lift_ntt(c, pts + (size_t)e * N, N, ptn); /* :225-228 */
          orc_multiply_plain_ntt(c, m->sv + (sv_off + i) * ctw, ptn, temp + ti * ctw); /* :229 */
          if (ct_is_transparent(c, temp + ti * ctw)) m->transparent = 1;
          ++ti;
        }
      }
      free(ptn);
      free(pts);
      free(lower);
    }
    if (first) { /* database.cpp:238-247 */
      rcount = tcount;
      result = (uint64_t*)malloc(sizeof(uint64_t) * ctw * rcount);
      memcpy(result, temp, sizeof(uint64_t) * ctw * rcount);
      first = 0;
    } else {
      for (uint64_t j = 0; j < rcount; ++j) {
        orc_ct_add_inplace(c, result + j * ctw, temp + j * ctw);
        if (ct_is_transparent(c, result + j * ctw)) m->transparent = 1;
      }
    }
  }
  free(temp);
  for (uint64_t j = 0; j < rcount; ++j) orc_ct_ntt_inv(c, result + j * ctw); /* database.cpp:250-254 */
  *count = rcount;
  return result;
}

/* reference database.cpp:290-316 */
int orc_db_multiply(const orc_ctx* c, const uint64_t* db_ntt, uint64_t P, const uint32_t* dims, uint32_t nd,
                    uint64_t* sv, uint8_t* sv_is_ntt, uint64_t sv_count, uint64_t* out, uint64_t* out_count) {
  uint64_t dim_sum = 0;
  for (uint32_t i = 0; i < nd; ++i) dim_sum += dims[i];
  if (sv_count != dim_sum) return ORC_INVALID_ARGUMENT;
  dbm_t m = {c, db_ntt, P, 0, sv, sv_is_ntt, orc_expansion_ratio(c), 0};
  uint64_t count = 0;
  uint64_t* r = dbm_multiply(&m, dims, nd, 0, &count);
  if (r) memcpy(out, r, sizeof(uint64_t) * 2 * c->k * c->N * count);
  free(r);
  if (out_count) *out_count = count;
  return m.transparent ? ORC_INTERNAL : ORC_OK; /* database.cpp:313-315 */
}

/* reference server.cpp:173-195 without the (de)serialisation at either end */
int orc_process_query(const orc_ctx* c, const uint64_t* db_ntt, uint64_t P, const uint32_t* dims, uint32_t nd,
                      const uint64_t* query_cts, uint32_t num_query_cts, const uint64_t* const* galois_keys,
                      uint64_t* out, uint64_t* out_count) {
  uint64_t dim_sum = 0;
  for (uint32_t i = 0; i < nd; ++i) dim_sum += dims[i];
  const size_t ctw = (size_t)2 * c->k * c->N;
  uint64_t* sv = (uint64_t*)malloc(sizeof(uint64_t) * ctw * (dim_sum ? dim_sum : 1));
  uint8_t* flags = (uint8_t*)calloc(dim_sum ? dim_sum : 1, 1);
  int rc = orc_oblivious_expansion_multi(c, query_cts, num_query_cts, dim_sum, galois_keys, sv);
  if (rc == ORC_OK) rc = orc_db_multiply(c, db_ntt, P, dims, nd, sv, flags, dim_sum, out, out_count);
  free(flags);
  free(sv);
  return rc;
}

/* ----------------------------------------------------------- string encoder */

/* reference string_encoder.cpp:25-27 (left-to-right integer division) */
uint64_t orc_items_per_plaintext(uint32_t N, uint32_t bits, uint64_t item_size) {
  return (uint64_t)N * bits / item_size / 8;
}
/* reference string_encoder.cpp:29-31 */
uint64_t orc_max_bytes_per_plaintext(uint32_t N, uint32_t bits) { return (uint64_t)N * bits / 8; }

/* reference string_encoder.cpp:58-122: bytes packed MSB-first into
 * bits_per_coeff-bit coefficients, final partial coefficient left-aligned. */
int orc_string_encode(const uint8_t* bytes, uint64_t nbytes, uint32_t bits, uint32_t N, uint64_t* coeffs,
                      uint32_t* num_coeff) {
  uint64_t nc = (uint64_t)ceil((double)(nbytes * 8) / bits);
  if (nc > N) return ORC_INVALID_ARGUMENT;
  for (uint64_t i = 0; i < nc; ++i) coeffs[i] = 0;
  uint64_t ci = 0;
  uint32_t coeff_bits = bits;
  for (uint64_t b = 0; b < nbytes; ++b) {
    uint8_t ch = bytes[b];
    uint32_t remain = 8;
    while (remain > 0) {
      uint32_t n = coeff_bits < remain ? coeff_bits : remain;
      coeffs[ci] <<= n;
      coeffs[ci] |= (uint64_t)(ch >> (8 - n));
      ch = (uint8_t)(ch << n);
      coeff_bits -= n;
      remain -= n;
      if (coeff_bits == 0) {
        ++ci;
        coeff_bits = bits;
      }
    }
  }
  if (coeff_bits < bits && coeff_bits > 0) coeffs[ci] <<= coeff_bits; /* terminate() */
  if (num_coeff) *num_coeff = (uint32_t)nc;
  return ORC_OK;
}

/* reference string_encoder.cpp:124-158 (length > 0 form) */
int orc_string_decode(const uint64_t* coeffs, uint32_t coeff_count, uint32_t bits, uint64_t length,
                      uint64_t byte_offset, uint8_t* out) {
  if ((byte_offset + length) > ((uint64_t)coeff_count * bits / 8)) return ORC_INVALID_ARGUMENT;
  uint64_t start = byte_offset * 8 / bits;
  uint64_t coeff_bits = ((start + 1) * bits) - (byte_offset * 8);
  memset(out, 0, length);
  uint64_t ri = 0;
  uint32_t remain = 8;
  for (uint64_t i = start; i < coeff_count; ++i) {
    while (coeff_bits > 0) {
      uint64_t n = coeff_bits < remain ? coeff_bits : remain;
      out[ri] = (uint8_t)(out[ri] << n);
      out[ri] |= (uint8_t)((coeffs[i] >> (coeff_bits - n)) & ((1u << n) - 1));
      coeff_bits -= n;
      remain -= (uint32_t)n;
      if (remain == 0) {
        if (++ri >= length) return ORC_OK;
        remain = 8;
      }
    }
    coeff_bits = bits;
  }
  return ORC_OK;
}

/* reference database.cpp:84-110 */
int orc_db_encode(const orc_ctx* c, const uint8_t* items, uint64_t num_items, uint64_t bytes_per_item,
                  uint64_t items_per_pt, uint32_t bits, uint64_t* db_ntt, uint64_t num_pt) {
  const uint32_t N = c->N;
  uint64_t* coeffs = (uint64_t*)malloc(sizeof(uint64_t) * N);
  if (bits == 0) bits = orc_bits_per_coeff(c->t);
  for (uint64_t i = 0; i < num_pt; ++i) {
    uint64_t first = i * items_per_pt;
    uint64_t cnt = first >= num_items ? 0 : (num_items - first < items_per_pt ? num_items - first : items_per_pt);
    uint32_t nc = 0;
    int rc = orc_string_encode(items + first * bytes_per_item, cnt * bytes_per_item, bits, N, coeffs, &nc);
    if (rc != ORC_OK) {
      free(coeffs);
      return rc;
    }
    orc_plain_lift_ntt(c, coeffs, nc, db_ntt + i * (size_t)c->k * N);
  }
  free(coeffs);
  return ORC_OK;
}

/* ------------------------------------------------------------ index math */

/* reference database.cpp:334-342 */
void orc_calculate_dimensions(uint32_t db_size, uint32_t num_dimensions, uint32_t* out) {
  uint32_t n = 0;
  for (int i = (int)num_dimensions; i > 0; --i) {
    out[n] = (uint32_t)ceil(pow((double)db_size, 1.0 / i));
    db_size = (uint32_t)ceil((double)db_size / out[n]);
    ++n;
  }
}

/* reference database.cpp:318-326 */
void orc_calculate_indices(uint32_t index, uint32_t items_per_pt, const uint32_t* dims, uint32_t nd, uint32_t* out) {
  uint32_t pt_index = index / items_per_pt;
  for (int i = (int)nd - 1; i >= 0; --i) {
    out[i] = pt_index % dims[i];
    pt_index = pt_index / dims[i];
  }
}

/* reference database.cpp:328-332 */
uint64_t orc_calculate_item_offset(uint32_t index, uint32_t items_per_pt, uint32_t bytes_per_item) {
  uint32_t pt_index = index / items_per_pt;
  return (uint64_t)(index - pt_index * items_per_pt) * bytes_per_item;
}
