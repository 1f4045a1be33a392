// host_math.h -- host-side number theory needed to build the device tables.
// (Product code: independent of oracle/.)
#pragma once
#include <stdint.h>

#include <cmath>

namespace pirgpu {
namespace hm {

typedef unsigned __int128 u128;

inline uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }

inline uint64_t powmod(uint64_t a, uint64_t e, uint64_t q) {
  uint64_t r = 1 % q;
  a %= q;
  while (e) {
    if (e & 1) r = mulmod(r, a, q);
    a = mulmod(a, a, q);
    e >>= 1;
  }
  return r;
}

inline uint64_t invmod_prime(uint64_t a, uint64_t q) { return powmod(a, q - 2, q); }

inline bool is_prime(uint64_t n) {
  if (n < 2) return false;
  const uint64_t bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  for (uint64_t b : bases) {
    if (n == b) return true;
    if (n % b == 0) return false;
  }
  uint64_t d = n - 1;
  int r = 0;
  while (!(d & 1)) {
    d >>= 1;
    ++r;
  }
  for (uint64_t b : bases) {
    uint64_t x = powmod(b, d, n);
    if (x == 1 || x == n - 1) continue;
    bool composite = true;
    for (int i = 1; i < r; ++i) {
      x = mulmod(x, x, n);
      if (x == n - 1) {
        composite = false;
        break;
      }
    }
    if (composite) return false;
  }
  return true;
}

// Smallest primitive 2N-th root of unity mod q -- the root SEAL's NTT tables use
// (util::try_minimal_primitive_root).  0 if none exists.
inline uint64_t minimal_primitive_root(uint64_t two_n, uint64_t q) {
  if ((q - 1) % two_n) return 0;
  const uint64_t e = (q - 1) / two_n;
  uint64_t root = 0;
  for (uint64_t x = 2; x < q && x < 100000; ++x) {
    uint64_t r = powmod(x, e, q);
    if (powmod(r, two_n / 2, q) == q - 1) {
      root = r;
      break;
    }
  }
  if (!root) return 0;
  const uint64_t sq = mulmod(root, root, q);
  uint64_t cur = root, best = root;
  for (uint64_t i = 0; i < two_n / 2; ++i) {
    if (cur < best) best = cur;
    cur = mulmod(cur, sq, q);
  }
  return best;
}

inline uint64_t shoup(uint64_t w, uint64_t q) { return (uint64_t)(((u128)w << 64) / q); }

inline uint32_t bitrev(uint32_t x, uint32_t bits) {
  uint32_t r = 0;
  for (uint32_t i = 0; i < bits; ++i) {
    r = (r << 1) | (x & 1);
    x >>= 1;
  }
  return r;
}

inline uint32_t ceil_log2(uint32_t v) {  // reference utils.cpp:30-44
  uint32_t r = 0;
  while ((1ull << r) < v) ++r;
  return r;
}

inline uint64_t next_power_two(uint64_t n) {  // reference utils.h:29-37
  uint64_t p = 1;
  while (p < n) p <<= 1;
  return p;
}

// bits per plaintext coefficient: <cmath> log2 of t truncated, as the reference
// computes it (ct_reencoder.cpp:32, string_encoder.cpp:85).
inline uint32_t bits_per_coeff(uint64_t t) { return (uint32_t)std::log2((double)t); }

// reference ct_reencoder.cpp:33-36,54-56
inline uint32_t local_expansion_ratio(uint64_t q, uint32_t b) {
  double bits = std::log2((double)q);
  return (uint32_t)std::ceil(bits / b);
}

}  // namespace hm
}  // namespace pirgpu
