// arith.h -- device-side modular arithmetic for gfx950 (integer and exact-fp64 flavours).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_params.h"

namespace pirgpu {

typedef unsigned __int128 u128;
typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------ integer

__device__ __forceinline__ uint64_t mul_shoup(uint64_t x, uint64_t w, uint64_t ws, uint64_t q) {
  uint64_t h = __umul64hi(x, ws);
  uint64_t r = x * w - h * q;
  return r >= q ? r - q : r;
}

// x * w mod q up to one multiple of q: result in [0, 2q) for any 64-bit x.
__device__ __forceinline__ uint64_t mul_shoup_lazy(uint64_t x, const Twiddle& t, uint64_t q) {
  return x * t.w - __umul64hi(x, t.ws) * q;
}

__device__ __forceinline__ uint64_t add_mod(uint64_t a, uint64_t b, uint64_t q) {
  uint64_t s = a + b;
  return s >= q ? s - q : s;
}

__device__ __forceinline__ uint64_t sub_mod(uint64_t a, uint64_t b, uint64_t q) {
  return a >= b ? a - b : a + q - b;
}

__device__ __forceinline__ uint64_t neg_mod(uint64_t a, uint64_t q) { return a ? q - a : 0; }

// x mod q for any 64-bit x (Barrett with floor(2^64 / q) = br_hi).
__device__ __forceinline__ uint64_t reduce64(uint64_t x, const ModConst& m) {
  uint64_t h = __umul64hi(x, m.br_hi);
  uint64_t r = x - h * m.q;
  return r >= m.q ? r - m.q : r;
}

// (hi:lo) mod q for any 128-bit input (SEAL barrett_reduce_128).
__device__ __forceinline__ uint64_t reduce128(uint64_t lo, uint64_t hi, const ModConst& m) {
  uint64_t carry = __umul64hi(lo, m.br_lo);
  uint64_t t2lo = lo * m.br_hi, t2hi = __umul64hi(lo, m.br_hi);
  uint64_t t1 = t2lo + carry;
  uint64_t t3 = t2hi + (t1 < t2lo);
  t2lo = hi * m.br_lo;
  t2hi = __umul64hi(hi, m.br_lo);
  uint64_t t1b = t1 + t2lo;
  carry = t2hi + (t1b < t1);
  uint64_t qhat = hi * m.br_hi + t3 + carry;
  uint64_t r = lo - qhat * m.q;
  return r >= m.q ? r - m.q : r;
}

__device__ __forceinline__ uint64_t mul_mod(uint64_t a, uint64_t b, const ModConst& m) {
  return reduce128(a * b, __umul64hi(a, b), m);
}

// ------------------------------------------------------------------ exact fp64
//
// gfx950 issues v_fma_f64 at the rate of v_mad_u64_u32 (profiles/r01_ubench_*), and an
// exact modular product costs 6 fp64 ops against ~25 integer ops, so for moduli
// below 2^49 the NTT kernels keep residues as doubles holding exact integers:
//   h = y*w (rounded), l = fma(y, w, -h) (exact error term), k = rint(h / q),
//   r = fma(-k, q, h) + l  ==  y*w - k*q   exactly, |r| <= (1/2 + eps) q.
// Every intermediate is an integer of magnitude < 2^53, so no rounding ever
// happens in the value that is kept; the result is a signed representative that
// is made canonical (and hence bit-identical to the integer path) at the end.

#pragma clang fp contract(off)

struct F64Mod {
  double q, qinv;
  bool lazy_inv = false;  // log2(q) + log2(N) <= 52: an inverse transform needs no intermediate renormalisation
};

__device__ __forceinline__ double f64_mulmod(double y, double w, const F64Mod& m) {
  const double h = y * w;
  const double l = __builtin_fma(y, w, -h);
  const double k = __builtin_rint(h * m.qinv);
  return __builtin_fma(-k, m.q, h) + l;
}

// signed representative of x mod q with |r| <= (1/2 + eps) q; exact for |x| < 2^53
__device__ __forceinline__ double f64_norm(double x, const F64Mod& m) {
  return __builtin_fma(-__builtin_rint(x * m.qinv), m.q, x);
}

__device__ __forceinline__ double f64_canon(double r, const F64Mod& m) { return r < 0.0 ? r + m.q : r; }

// exact for v < 2^53
__device__ __forceinline__ double f64_from_u64(uint64_t v) {
  return __builtin_fma((double)(uint32_t)(v >> 32), 4294967296.0, (double)(uint32_t)v);
}

// r a non-negative integer < 2^53
__device__ __forceinline__ uint64_t f64_to_u64(double r) {
  const uint32_t hi = (uint32_t)(r * 2.3283064365386962890625e-10);  // trunc(r / 2^32)
  const uint32_t lo = (uint32_t)__builtin_fma(-(double)hi, 4294967296.0, r);
  return ((uint64_t)hi << 32) | lo;
}

// ---- 40-bit packed storage of signed representatives (fp64 flavours, every modulus < 2^39) ----
//
// A key-switch intermediate x (an exact integer, |x| <= q) is kept in HBM as the unsigned integer u = x + q
// < 2^40: 4 low bytes + 1 high byte (5 N bytes per polynomial instead of 8 N).  Adding the magic constant
// 2^52 + q to the double puts u into the low mantissa bits, so packing is ONE v_add_f64 (the low word and the
// low byte of the high word are stored as they are) and unpacking is one v_or + one v_add_f64 -- against ~10
// conversion instructions per element for a canonical u64 round trip.
__device__ __forceinline__ double f64_pack_magic(double q) { return 4503599627370496.0 + q; }  // 2^52 + q

__device__ __forceinline__ void f64_pack40(double x, double magic, uint32_t& lo, uint32_t& hi) {
  const uint64_t b = (uint64_t)__double_as_longlong(x + magic);
  lo = (uint32_t)b;
  hi = (uint32_t)(b >> 32);  // stored with a byte store: only bits 32..39 of u
}

__device__ __forceinline__ double f64_unpack40(uint32_t lo, uint32_t hi_byte, double magic) {
  return __longlong_as_double((long long)(((uint64_t)(0x43300000u | hi_byte) << 32) | lo)) - magic;
}

}  // namespace pirgpu
