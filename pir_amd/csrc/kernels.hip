// kernels.hip -- gfx950 (MI355X) kernels of the PIR server query path.
//
// Integer modular arithmetic over 64-bit RNS residues: no MFMA, the bounds are
// HBM bandwidth (database scan) and integer-multiply issue rate (NTT).  Every
// kernel writes canonical residues in [0, q), so results are bit-identical to
// the reference's SEAL CPU path.
//
// Reference call sites each kernel replaces (paths relative to /root/reference):
//   ntt_batch_kernel          Evaluator::transform_to/from_ntt_inplace   database.cpp:190,222,252
//   db_encode_kernel          StringEncoder::encode + transform_to_ntt   database.cpp:100-106, string_encoder.cpp:58-122
//   ks_main_kernel,
//   ks_combine_kernel         one level of oblivious_expansion           server.cpp:120-142 (apply_galois_inplace :71,
//                                                                         negacyclic_shift :97, add_inplace :140-141)
//   scan_kernel               multiply_plain + add_inplace base case     database.cpp:185-194,238-247
//   scan_mq_kernel            the same, 1/2/4 queries per database pass, selectors shared through LDS
//   reduce_splits_kernel      add_inplace of partial sums (column splits, chunk sums, multi-GPU fix-up)
// (kernels that contain an NTT live in ntt_kernels.hip)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>

#include <type_traits>

#include "arith.h"
#include "device_params.h"
#include "kernels.h"

namespace pirgpu {

// SEAL NTT order <-> device NTT order for npolys polynomials (boundary only:
// Galois-key upload and the test hooks).  to_device: out[e*NT + t] = in[EPT t + e], EPT = 2^ntt_log_ept(logN).
// as_f64: the device-order side holds the residues as exact doubles (fp64 NTT flavours).
__global__ void ntt_reorder_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint32_t logN,
                                   uint64_t npolys, int to_device, int as_f64) {
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t N = 1u << logN, le = (uint32_t)ntt_log_ept((int)logN), NT = N >> le;
  if (gid >= npolys << logN) return;
  const uint64_t poly = gid >> logN;
  const uint32_t i = (uint32_t)(gid & (N - 1));  // device slot
  const uint32_t e = i / NT, t = i % NT;
  const uint32_t seal = (t << le) | e;
  if (to_device) {
    uint64_t v = in[(poly << logN) + seal];
    if (as_f64) v = (uint64_t)__double_as_longlong(f64_from_u64(v));
    out[(poly << logN) + i] = v;
  } else {
    uint64_t v = in[(poly << logN) + i];
    if (as_f64) v = f64_to_u64(__longlong_as_double((long long)v));
    out[(poly << logN) + seal] = v;
  }
}

// ------------------------------------------------------------------ expansion

// Part 2: divide-and-round by the special prime, add sigma_g(c0), then the tree
// butterfly   res_out[n] = a + g,   res_out[n + nodes] = x^(-2^j) * (a - g)
// (identical residues to the reference's two negacyclic shifts + adds, since
// x^-(N+2^j) = -x^(-2^j)).  With expand_step == 0 only g is written (plain
// substitute_power_x_inplace).  One thread per (node, residue, coefficient).
template <bool P40>
__global__ void ks_combine_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ res_in,
                                  const uint64_t* __restrict__ prod, uint32_t galois_inv, uint32_t nodes,
                                  uint32_t shift_pow, int expand_step, uint32_t hi_limit,
                                  uint64_t* __restrict__ res_out) {
  const uint32_t N = P->N, k = P->k, km = k + 1;
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t total = (uint64_t)nodes * k * N;
  if (gid >= total) return;
  const uint32_t i = (uint32_t)(gid & (N - 1));
  const uint32_t j = (uint32_t)((gid >> P->logN) % k);
  const uint32_t node = (uint32_t)(gid / ((uint64_t)k * N));
  const ModConst mj = P->mod[j];
  const ModConst mp = P->mod[k];
  const uint64_t q = mj.q;
  // gather index for sigma_g(c0)[i]
  uint32_t raw = (i * galois_inv) & (2 * N - 1);
  uint32_t src_i = raw & (N - 1);
  bool neg = raw >= N;
  uint64_t g[2];
#pragma unroll
  for (int comp = 0; comp < 2; ++comp) {
    uint64_t sp, dj;  // special-prime and data-prime residues of the key-switch product
    if constexpr (P40) {  // 5-byte polynomials: N low words, then N high bytes (ntt_kernels.hip load40/store40)
      const uint8_t* pr = reinterpret_cast<const uint8_t*>(prod) + ((size_t)node * 2 + comp) * km * 5 * N;
      const uint8_t* ps = pr + (size_t)k * 5 * N;
      const uint8_t* pj = pr + (size_t)j * 5 * N;
      sp = (uint64_t)reinterpret_cast<const uint32_t*>(ps)[i] | ((uint64_t)ps[4 * (size_t)N + i] << 32);
      dj = (uint64_t)reinterpret_cast<const uint32_t*>(pj)[i] | ((uint64_t)pj[4 * (size_t)N + i] << 32);
    } else {
      const uint64_t* pr = prod + ((size_t)node * 2 + comp) * km * N;
      sp = pr[(size_t)k * N + i];
      dj = pr[(size_t)j * N + i];
    }
    uint64_t r = add_mod(sp, P->p_half, mp.q);
    uint64_t delta = sub_mod(reduce64(r, mj), P->p_half_mod[j], q);
    uint64_t v = sub_mod(dj, delta, q);
    g[comp] = mul_shoup(v, P->p_inv[j], P->p_inv_s[j], q);
  }
  const uint64_t* a_ct = res_in + (size_t)node * 2 * k * N;
  uint64_t c0 = a_ct[(size_t)j * N + src_i];
  if (neg) c0 = neg_mod(c0, q);
  g[0] = add_mod(g[0], c0, q);
  if (!expand_step) {
    uint64_t* o = res_out + (size_t)node * 2 * k * N;
    o[(size_t)j * N + i] = g[0];
    o[((size_t)k + j) * N + i] = g[1];
    return;
  }
  // x^(-2^j): negacyclic shift by 2N - shift_pow
  uint32_t sraw = i + (2 * N - shift_pow);
  uint32_t sidx = sraw & (N - 1);
  bool sneg = (sraw & N) != 0;
  uint64_t* lo = res_out + (size_t)node * 2 * k * N;
  uint64_t* hi = res_out + ((size_t)node + nodes) * 2 * k * N;
  const bool want_hi = node + nodes < hi_limit;  // last level: outputs past the requested count are never read
#pragma unroll
  for (int comp = 0; comp < 2; ++comp) {
    size_t off = ((size_t)comp * k + j) * N;
    uint64_t a = a_ct[off + i];
    lo[off + i] = add_mod(a, g[comp], q);
    if (!want_hi) continue;
    uint64_t d = sub_mod(a, g[comp], q);
    if (sneg) d = neg_mod(d, q);
    hi[off + sidx] = d;
  }
}

// The same step for the fp64 flavours.  The expansion tree is held as doubles (exact integers, signed
// representatives |x| <= (1/2 + eps) q_j), the key-switch products arrive as signed representatives as well
// (offset 40-bit packed, arith.h f64_pack40, or plain doubles), and everything is fp64 arithmetic: ~100 VALU
// instructions per element against ~325 for the 64-bit integer version.  Divide-and-round by the special prime
// (SURVEY App. A.4): with r = (s + floor(p/2)) mod p, r - floor(p/2) is the CENTRED representative c of the
// special-prime residue s, so delta_j = c mod q_j; c must be exactly centred (a signed representative may be
// off-centre by eps p, which would change delta by p mod q_j), the data residue may be any representative.
// PB: bytes per residue of the packed products (0: plain doubles; 5 / 6 / 7: ntt_kernels.hip store40f at that width)
template <int PB>
__global__ void ks_combine_f64_kernel(const DevParams* __restrict__ P, const double* __restrict__ res_in,
                                      const uint64_t* __restrict__ prod, uint32_t galois_inv, uint32_t nodes,
                                      uint32_t shift_pow, int expand_step, uint32_t hi_limit,
                                      double* __restrict__ res_out) {
  const uint32_t N = P->N, k = P->k, km = k + 1;
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t total = (uint64_t)nodes * k * N;
  if (gid >= total) return;
  const uint32_t i = (uint32_t)(gid & (N - 1));
  const uint32_t j = (uint32_t)((gid >> P->logN) % k);
  const uint32_t node = (uint32_t)(gid / ((uint64_t)k * N));
  const F64Mod mj{P->tab[j].qd, P->tab[j].qinvd};
  const double pf = P->p_f, half = P->p_half_f, pinv = P->p_inv_f[j];
  const uint32_t raw = (i * galois_inv) & (2 * N - 1);  // gather index for sigma_g(c0)[i]
  const uint32_t src_i = raw & (N - 1);
  const bool neg = raw >= N;
  double g[2];
#pragma unroll
  for (int comp = 0; comp < 2; ++comp) {
    double sp, dj;  // special-prime and data-prime residues of the key-switch product (signed representatives)
    if constexpr (PB != 0) {
      const uint8_t* pr = reinterpret_cast<const uint8_t*>(prod) + ((size_t)node * 2 + comp) * km * PB * N;
      const uint8_t* ps = pr + (size_t)k * PB * N;
      const uint8_t* pj = pr + (size_t)j * PB * N;
      // high bytes thread-major (ntt_kernels.hip store40f): the PB - 4 bytes of element e * (N/EPT) + t at 4 N + (PB - 4) (EPT t + e)
      const uint32_t le = (uint32_t)ntt_log_ept((int)P->logN);
      const size_t hi_at = 4 * (size_t)N + (PB - 4) * ((((size_t)(i & ((N >> le) - 1))) << le) + (i >> (P->logN - le)));
      uint32_t hs = 0, hj = 0;
#pragma unroll
      for (int b = PB - 5; b >= 0; --b) {
        hs = (hs << 8) | ps[hi_at + b];
        hj = (hj << 8) | pj[hi_at + b];
      }
      // (the 7-byte form's third byte already carries the 0x30 of the double's exponent: the OR inside f64_unpack40 is idempotent)
      sp = f64_unpack40(reinterpret_cast<const uint32_t*>(ps)[i], hs, f64_pack_magic(pf));
      dj = f64_unpack40(reinterpret_cast<const uint32_t*>(pj)[i], hj, f64_pack_magic(mj.q));
    } else {
      const double* pr = reinterpret_cast<const double*>(prod) + ((size_t)node * 2 + comp) * km * N;
      sp = pr[(size_t)k * N + i];
      dj = pr[(size_t)j * N + i];
    }
    sp = sp > half ? sp - pf : sp;   // exact centring: c in [-(p-1)/2, (p-1)/2]
    sp = sp < -half ? sp + pf : sp;
    const double delta = f64_norm(sp, mj);
    g[comp] = f64_mulmod(dj - delta, pinv, mj);
  }
  const double* a_ct = res_in + (size_t)node * 2 * k * N;
  const double c0 = a_ct[(size_t)j * N + src_i];
  g[0] += neg ? -c0 : c0;
  if (!expand_step) {
    double* o = res_out + (size_t)node * 2 * k * N;
    o[(size_t)j * N + i] = f64_norm(g[0], mj);
    o[((size_t)k + j) * N + i] = g[1];
    return;
  }
  const uint32_t sraw = i + (2 * N - shift_pow);  // x^(-2^j): negacyclic shift by 2N - shift_pow
  const uint32_t sidx = sraw & (N - 1);
  const bool sneg = (sraw & N) != 0;
  double* lo = res_out + (size_t)node * 2 * k * N;
  double* hi = res_out + ((size_t)node + nodes) * 2 * k * N;
  const bool want_hi = node + nodes < hi_limit;
#pragma unroll
  for (int comp = 0; comp < 2; ++comp) {
    const size_t off = ((size_t)comp * k + j) * N;
    const double a = a_ct[off + i];
    lo[off + i] = f64_norm(a + g[comp], mj);
    if (!want_hi) continue;
    const double d = f64_norm(a - g[comp], mj);
    hi[off + sidx] = sneg ? -d : d;
  }
}

// Expansion tree <-> canonical u64 ciphertexts for the fp64 flavours (query import, test-hook export).
// chunk_words / in_stride: the input is a sequence of chunks of chunk_words words, in_stride words apart (the B query
// ciphertexts of a group inside the staged batch); contiguous input: in_stride == chunk_words.
__global__ void tree_import_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ in,
                                   double* __restrict__ out, uint64_t words, uint64_t chunk_words, uint64_t in_stride) {
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= words) return;
  const uint32_t j = (uint32_t)((gid >> P->logN) % P->k);
  const uint64_t chunk = gid / chunk_words;
  out[gid] = f64_norm(f64_from_u64(in[chunk * in_stride + (gid - chunk * chunk_words)]), F64Mod{P->tab[j].qd, P->tab[j].qinvd});
}

__global__ void tree_export_kernel(const DevParams* __restrict__ P, const double* __restrict__ in,
                                   uint64_t* __restrict__ out, uint64_t words) {
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= words) return;
  const uint32_t j = (uint32_t)((gid >> P->logN) % P->k);
  const F64Mod m{P->tab[j].qd, P->tab[j].qinvd};
  out[gid] = f64_to_u64(f64_canon(f64_norm(in[gid], m), m));
}

// Split upper level, part 2 (see upper_ntt_kernel): products of the transformed plaintexts of one block of children
// with both selector polynomials, summed over the block.  One thread per (query, row, cc, GROUP of EG Encode chunks,
// target modulus, slot i): the two selector words of a child are loaded once and multiplied into the EG chunks'
// transformed plaintexts (one thread per chunk re-read the selectors E times -- at cfg 5, E = 24, that was 17 x the
// selectors' bytes from HBM and more than the transformed plaintexts themselves: 7.4 GB per launch against 3.3).
// Sums are signed representatives kept as doubles in `acc` between blocks; the last block writes canonical residues
// (u64) where reduce_splits_kernel would have.
template <int EG>
__global__ void __launch_bounds__(256)
upper_mac_kernel(const DevParams* __restrict__ P, const double* __restrict__ scratch, MfmaPtrs svq,
                 double* __restrict__ acc_all, uint64_t* __restrict__ out_all, uint32_t n_rows, uint32_t C,
                 uint32_t E, uint32_t sv_first, uint32_t b0, uint32_t blk, uint32_t n_dim, int first,
                 int last, uint64_t acc_qstride, uint64_t out_qstride) {
  const uint32_t N = P->N, k = P->k;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;   // slot
  const uint32_t jt = blockIdx.y % k, e0 = (blockIdx.y / k) * EG;
  const uint32_t cc = blockIdx.z % C, r = (blockIdx.z / C) % n_rows, qi = blockIdx.z / (C * n_rows);
  if (i >= N) return;
  const F64Mod m{P->tab[jt].qd, P->tab[jt].qinvd};
  const uint64_t* sv = reinterpret_cast<const uint64_t*>(svq.p[qi]);
  const size_t chunk_stride = (size_t)k * N;                    // scratch is [child in block][chunk][target modulus][N]
  const double* x = scratch + ((((size_t)qi * n_rows + r) * C + cc) * blk * E + e0) * chunk_stride + (size_t)jt * N + i;
  const size_t child_stride = (size_t)E * chunk_stride;
  double a0[EG], a1[EG];
#pragma unroll
  for (int g = 0; g < EG; ++g) a0[g] = a1[g] = 0.0;
  uint32_t since = 0;
  const uint32_t n_here = b0 < n_dim ? (n_dim - b0 < blk ? n_dim - b0 : blk) : 0;
  for (uint32_t iib = 0; iib < n_here; ++iib) {
    const uint64_t* s0 = sv + (((size_t)(sv_first + b0 + iib) * 2 + 0) * k + jt) * N + i;
    const double w0 = f64_from_u64(s0[0]), w1 = f64_from_u64(s0[(size_t)k * N]);
    double v[EG];
#pragma unroll
    for (int g = 0; g < EG; ++g) v[g] = x[(size_t)iib * child_stride + (size_t)g * chunk_stride];
#pragma unroll
    for (int g = 0; g < EG; ++g) {
      a0[g] += f64_mulmod(v[g], w0, m);
      a1[g] += f64_mulmod(v[g], w1, m);
    }
    if (++since == 8) {
      since = 0;
#pragma unroll
      for (int g = 0; g < EG; ++g) {
        a0[g] = f64_norm(a0[g], m);
        a1[g] = f64_norm(a1[g], m);
      }
    }
  }
  double* acc = acc_all + (size_t)qi * acc_qstride;
  uint64_t* out = out_all + (size_t)qi * out_qstride;
#pragma unroll
  for (int g = 0; g < EG; ++g) {
    const size_t slot = (((size_t)r * C + cc) * E + e0 + g);
    const size_t o0 = ((slot * 2 + 0) * k + jt) * N + i, o1 = ((slot * 2 + 1) * k + jt) * N + i;
    double s0 = a0[g], s1 = a1[g];
    if (!first) {
      s0 += acc[o0];
      s1 += acc[o1];
    }
    s0 = f64_norm(s0, m);
    s1 = f64_norm(s1, m);
    if (last) {
      out[o0] = f64_to_u64(f64_canon(s0, m));
      out[o1] = f64_to_u64(f64_canon(s1, m));
    } else {
      acc[o0] = s0;
      acc[o1] = s1;
    }
  }
}

// multiply_inverse_power_of_x on whole ciphertexts (reference server.cpp:78-103).
__global__ void monomial_shift_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ in,
                                      uint32_t shift, uint64_t count, uint64_t* __restrict__ out) {
  const uint32_t N = P->N, k = P->k;
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= count * 2 * k * N) return;
  const uint32_t i = (uint32_t)(gid & (N - 1));
  const uint64_t poly = gid >> P->logN;
  const uint32_t j = (uint32_t)(poly % k);
  const uint64_t q = P->mod[j].q;
  uint64_t v = in[gid];
  if (shift == 0) {
    out[gid] = v;
    return;
  }
  uint32_t raw = i + shift;
  if ((raw & N) && v) v = q - v;
  out[poly * N + (raw & (N - 1))] = v;
}

// ------------------------------------------------------------------ database scan

// VEC adjacent residues with one (16-byte when VEC == 2) global load.
template <int VEC>
__device__ __forceinline__ void load_vec(const uint64_t* __restrict__ p, uint64_t (&d)[VEC]) {
  if constexpr (VEC == 2) {
    u64x2 v = *reinterpret_cast<const u64x2*>(p);
    d[0] = v.x;
    d[1] = v.y;
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) d[i] = p[i];
  }
}

// Base case of DatabaseMultiplier::multiply fused over whole rows (the streaming kernel):
//   out[split][row][p][j][c] = sum_{col in split} sv[col][p][j][c] * db[row*cols + col][j][c]  mod q_j
// Each thread owns VEC adjacent residues for ROWS rows and streams the database
// exactly once with 16-byte non-temporal loads (fully coalesced, 1 KiB per wave
// instruction); the selectors are re-read from L2 (one load per ROWS database
// loads).  Loads of column c+1 are issued before the products of column c are
// formed (two register buffers), so every wave always has (ROWS+2) x 1 KiB in
// flight.  Products accumulate lazily in 128 bits and are reduced once (or every
// lazy_limit columns for wide moduli).
// grid = (k*N / (VEC*block), ceil(rows / ROWS), nsplit).
template <int ROWS, int VEC>
struct ScanBuf {
  uint64_t d[ROWS][VEC];
  uint64_t s[2][VEC];
};

template <int ROWS, int VEC>
__device__ __forceinline__ void scan_load(ScanBuf<ROWS, VEC>& b, const uint64_t* const (&rp)[ROWS],
                                          const uint64_t* __restrict__ svp, uint32_t col, uint32_t kN) {
  const size_t off = (size_t)col * kN;
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    if constexpr (VEC == 2) {
      u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(rp[r] + off));
      b.d[r][0] = v.x;
      b.d[r][1] = v.y;
    } else {
#pragma unroll
      for (int v = 0; v < VEC; ++v) b.d[r][v] = __builtin_nontemporal_load(rp[r] + off + v);
    }
  }
  load_vec<VEC>(svp + 2 * off, b.s[0]);
  load_vec<VEC>(svp + 2 * off + kN, b.s[1]);
}

// Lazy accumulators of residue products.
//  AccWide: exact 128-bit sum, any modulus < 2^61, up to lazy_limit terms.
//  AccLimb: for moduli < 2^50 the residues are split at bit 28 and the four
//    partial products accumulate in three 64-bit sums with no carry handling at
//    all (4 v_mad_u64_u32 per product, ~3x fewer VALU ops than the 128-bit path);
//    exact for up to kLimbLazy terms: s00 < 128 * 2^56, s01 < 128 * 2^51, s11 < 128 * 2^44.
constexpr uint32_t kLimbLazy = 128;

struct AccWide {
  u128 v;
  __device__ __forceinline__ void clear() { v = 0; }
  __device__ __forceinline__ uint64_t fold(const ModConst& m) const {
    return reduce128((uint64_t)v, (uint64_t)(v >> 64), m);
  }
  __device__ __forceinline__ void set(uint64_t r) { v = r; }
};

struct AccLimb {
  uint64_t s00, s01, s11;
  __device__ __forceinline__ void clear() { s00 = s01 = s11 = 0; }
  __device__ __forceinline__ uint64_t fold(const ModConst& m) const {
    u128 t = (u128)s00 + ((u128)s01 << 28) + ((u128)s11 << 56);
    return reduce128((uint64_t)t, (uint64_t)(t >> 64), m);
  }
  __device__ __forceinline__ void set(uint64_t r) {
    s00 = r;
    s01 = s11 = 0;
  }
};

template <int ROWS, int VEC>
__device__ __forceinline__ void scan_mac(AccWide (&acc)[ROWS][2][VEC], const ScanBuf<ROWS, VEC>& b) {
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      acc[r][0][v].v += (u128)b.s[0][v] * b.d[r][v];
      acc[r][1][v].v += (u128)b.s[1][v] * b.d[r][v];
    }
}

template <int ROWS, int VEC>
__device__ __forceinline__ void scan_mac(AccLimb (&acc)[ROWS][2][VEC], const ScanBuf<ROWS, VEC>& b) {
  uint32_t a0[2][VEC], a1[2][VEC];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      a0[p][v] = (uint32_t)b.s[p][v] & 0x0FFFFFFFu;
      a1[p][v] = (uint32_t)(b.s[p][v] >> 28);
    }
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const uint32_t b0 = (uint32_t)b.d[r][v] & 0x0FFFFFFFu;
      const uint32_t b1 = (uint32_t)(b.d[r][v] >> 28);
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        acc[r][p][v].s00 += (uint64_t)a0[p][v] * b0;
        acc[r][p][v].s01 += (uint64_t)a0[p][v] * b1;
        acc[r][p][v].s01 += (uint64_t)a1[p][v] * b0;
        acc[r][p][v].s11 += (uint64_t)a1[p][v] * b1;
      }
    }
}

template <int ROWS, int VEC, typename ACC>
__global__ void __launch_bounds__(256)
scan_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ db, const uint64_t* __restrict__ sv,
            uint64_t* __restrict__ out, uint32_t rows, uint32_t cols, uint64_t num_pt, uint32_t cols_per_split) {
  const uint32_t N = P->N, k = P->k;
  const uint32_t kN = k * N;
  const uint32_t c0 = (blockIdx.x * blockDim.x + threadIdx.x) * VEC;  // index into [k][N]
  if (c0 >= kN) return;
  const uint32_t j = c0 >> P->logN;
  const ModConst m = P->mod[j];
  const uint32_t row0 = blockIdx.y * ROWS;
  const uint32_t col_begin = blockIdx.z * cols_per_split;
  uint32_t col_end = col_begin + cols_per_split;
  if (col_end > cols) col_end = cols;
  const uint32_t lazy = std::is_same<ACC, AccLimb>::value ? kLimbLazy : P->lazy_limit;

  ACC acc[ROWS][2][VEC];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[r][p][v].clear();

  // Rows past the end of this launch are redirected to the last real row (their
  // sums are discarded); only the database's final row can be shorter than cols.
  const uint64_t* rp[ROWS];
  uint32_t ncol[ROWS];
  uint32_t nfull = col_end;
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    uint32_t row = row0 + r < rows ? row0 + r : rows - 1;
    uint64_t first = (uint64_t)row * cols;
    uint64_t avail = first < num_pt ? num_pt - first : 0;
    ncol[r] = avail > cols ? cols : (uint32_t)avail;
    if (ncol[r] < nfull) nfull = ncol[r];
    rp[r] = db + first * kN + c0;
  }
  if (nfull < col_begin) nfull = col_begin;
  const uint64_t* svp = sv + c0;

  // ---- main loop: columns [col_begin, nfull) exist in every row of the group
  for (uint32_t chunk = col_begin; chunk < nfull;) {
    const uint32_t chunk_end = nfull - chunk > lazy ? chunk + lazy : nfull;
    // Two register buffers; the prefetches are unconditional (the column index is
    // clamped) so the s_waitcnt counters stay exact and a full buffer is always in flight.
    ScanBuf<ROWS, VEC> A, B;
    const uint32_t last = chunk_end - 1;
    scan_load<ROWS, VEC>(A, rp, svp, chunk, kN);
    uint32_t col = chunk;
    for (; col + 2 <= chunk_end; col += 2) {
      scan_load<ROWS, VEC>(B, rp, svp, col + 1, kN);
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the products (hipcc sinks loads otherwise)
      scan_mac<ROWS, VEC>(acc, A);
      __builtin_amdgcn_sched_barrier(0);
      scan_load<ROWS, VEC>(A, rp, svp, col + 2 < last ? col + 2 : last, kN);
      __builtin_amdgcn_sched_barrier(0);
      scan_mac<ROWS, VEC>(acc, B);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (col < chunk_end) scan_mac<ROWS, VEC>(acc, A);  // odd count: A holds column `last`
    chunk = chunk_end;
    if (chunk < col_end) {  // more columns follow: fold the lazy sums
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[r][p][v].set(acc[r][p][v].fold(m));
    }
  }
  // ---- ragged tail (only the group holding the database's last, shorter row)
  uint32_t since = 0;
  for (uint32_t col = nfull; col < col_end; ++col) {
    ScanBuf<1, VEC> tb;
    load_vec<VEC>(svp + (size_t)col * 2 * kN, tb.s[0]);
    load_vec<VEC>(svp + (size_t)col * 2 * kN + kN, tb.s[1]);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      if (col < ncol[r]) {
        load_vec<VEC>(rp[r] + (size_t)col * kN, tb.d[0]);
        scan_mac<1, VEC>(reinterpret_cast<ACC(&)[1][2][VEC]>(acc[r]), tb);
      }
    }
    if (++since == lazy) {
      since = 0;
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[r][p][v].set(acc[r][p][v].fold(m));
    }
  }
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    uint32_t row = row0 + r;
    if (row < rows) {
      uint64_t* o = out + ((size_t)blockIdx.z * rows + row) * 2 * kN + c0;
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int v = 0; v < VEC; ++v) o[(size_t)p * kN + v] = acc[r][p][v].fold(m);
    }
  }
}

// AccLimb products with selectors that were already split at staging time (low limb in the
// low dword, high limb in the high dword of b.s).
template <int ROWS>
__device__ __forceinline__ void scan_mac_presplit(AccLimb (&acc)[ROWS][2][2], const ScanBuf<ROWS, 2>& b) {
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const uint32_t b0 = (uint32_t)b.d[r][v] & 0x0FFFFFFFu;
      const uint32_t b1 = (uint32_t)(b.d[r][v] >> 28);
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const uint32_t a0 = (uint32_t)b.s[p][v], a1 = (uint32_t)(b.s[p][v] >> 32);
        acc[r][p][v].s00 += (uint64_t)a0 * b0;
        acc[r][p][v].s01 += (uint64_t)a0 * b1;
        acc[r][p][v].s01 += (uint64_t)a1 * b0;
        acc[r][p][v].s11 += (uint64_t)a1 * b1;
      }
    }
}

// ------------------------------------------------------------------ multi-query scan
//
// Batch mode: NQ queries share one pass over the database, so each query pays 1/NQ of the
// HBM traffic.  A workgroup = 4 waves that own the SAME 128 residues (64 lanes x 2) of
// 4*ROWS_W consecutive rows; the NQ x 2 selector rows of those residues are staged once per
// workgroup in LDS (tiles of TCOLS columns, double buffered, loads for tile t+1 in flight
// while tile t is consumed) instead of being re-read from L2 by every wave.  Database rows
// are zero padded to full length in HBM (ctx.hip), so the loop is branch-free.
struct MqArgs {
  const uint64_t* sv[kMaxScanQueries];  // per query: first selector of the scanned dimension (NTT form)
  uint64_t* out[kMaxScanQueries];       // per query: [rows][2][k][N]
};

template <int ROWS_W, int NQ, int TCOLS, typename ACC, int MINWAVES = 1>
__global__ void __launch_bounds__(256, MINWAVES)
scan_mq_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ db, MqArgs a, uint32_t rows,
               uint32_t cols) {
  extern __shared__ __attribute__((aligned(16))) unsigned char mq_smem[];
  u64x2(*tile)[TCOLS * NQ * 2][64] = reinterpret_cast<u64x2(*)[TCOLS * NQ * 2][64]>(mq_smem);
  constexpr int PIECES = TCOLS * NQ * 2 * 64 / 256;
  const uint32_t N = P->N, k = P->k;
  const uint32_t kN = k * N;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware block order (workgroup b runs on XCD b % 8, each XCD has its own 4 MiB L2): an XCD
  // walks all row blocks of one residue chunk before it moves to the next chunk, so the
  // selector rows of only a few chunks are live in its L2 at any time.
  const uint32_t n_chunks = kN / 128, n_rb = (rows + 4 * ROWS_W - 1) / (4 * ROWS_W);
  uint32_t chunk, rb;
  if ((n_chunks & 7) == 0) {
    const uint32_t xcd = blockIdx.x & 7, t = blockIdx.x >> 3;
    chunk = xcd + 8 * (t / n_rb);
    rb = t % n_rb;
  } else {
    chunk = blockIdx.x % n_chunks;
    rb = blockIdx.x / n_chunks;
  }
  const uint32_t chunk_base = chunk * 128;
  const uint32_t c0 = chunk_base + lane * 2;
  const uint32_t j = c0 >> P->logN;
  const ModConst m = P->mod[j];
  const uint32_t row0 = rb * (4 * ROWS_W) + wave * ROWS_W;
  const uint32_t lazy = std::is_same<ACC, AccLimb>::value ? kLimbLazy : P->lazy_limit;

  const uint64_t* rp[ROWS_W];
#pragma unroll
  for (int r = 0; r < ROWS_W; ++r) {
    const uint32_t row = row0 + r < rows ? row0 + r : rows - 1;
    rp[r] = db + (size_t)row * cols * kN + c0;
  }
  ACC acc[NQ][ROWS_W][2][2];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int r = 0; r < ROWS_W; ++r)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        acc[q][r][p][0].clear();
        acc[q][r][p][1].clear();
      }

  const uint32_t ntiles = (cols + TCOLS - 1) / TCOLS;
  const uint32_t last_col = cols - 1;
  u64x2 st[PIECES];
  static_assert(TCOLS == 4, "the database ring below assumes 4-column tiles");
  // ring of database columns with prefetch distance RING/2 (static indices: t is unrolled and the
  // tile parity BUF is a template constant).  One row per wave needs the deeper ring to keep
  // enough bytes in flight (4 x 1 KiB per wave).
  constexpr int RING = ROWS_W == 1 ? 8 : 4, DIST = RING / 2;
  u64x2 dbuf[RING][ROWS_W];

  auto stage_load = [&](uint32_t t0) {
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const uint32_t piece = tid + 256u * i;
      const uint32_t ridx = piece >> 6, l = piece & 63;
      const uint32_t t = ridx / (NQ * 2), q = (ridx >> 1) % NQ, p = ridx & 1;
      const uint32_t col = t0 + t < last_col ? t0 + t : last_col;
      st[i] = *reinterpret_cast<const u64x2*>(a.sv[q] + (size_t)col * 2 * kN + (size_t)p * kN + chunk_base + 2 * l);
    }
  };
  // limb accumulators: the selectors are split at bit 28 once here (low limb in the low dword,
  // high limb in the high dword) instead of per wave and column in the inner loop
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const uint32_t piece = tid + 256u * i;
      u64x2 v = st[i];
      if constexpr (std::is_same<ACC, AccLimb>::value) {
        v.x = ((v.x >> 28) << 32) | (v.x & 0x0FFFFFFFull);
        v.y = ((v.y >> 28) << 32) | (v.y & 0x0FFFFFFFull);
      }
      tile[buf][piece >> 6][piece & 63] = v;
    }
  };
  uint32_t since = 0;
  // one tile step with compile-time buffer indices (runtime-indexed register arrays would
  // be demoted to scratch): consumes buffers BUF, prefetches the next tile into BUF^1
  auto step = [&](auto bufc, uint32_t tl) {
    constexpr int BUF = decltype(bufc)::value;
    const uint32_t t0 = tl * TCOLS;
    // prefetch the next selector tile (clamped, hence unconditional: exact s_waitcnt counters)
    stage_load(t0 + TCOLS);
#pragma unroll
    for (int t = 0; t < TCOLS; ++t) {
      {  // database column t+DIST -> its ring slot, DIST columns ahead of its use
        const uint32_t col = t0 + t + DIST < last_col ? t0 + t + DIST : last_col;
#pragma unroll
        for (int r = 0; r < ROWS_W; ++r)
          dbuf[(BUF * TCOLS + t + DIST) % RING][r] =
              __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(rp[r] + (size_t)col * kN));
      }
      __builtin_amdgcn_sched_barrier(0);
      if (t0 + t < cols) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          ScanBuf<ROWS_W, 2> sb;
          const u64x2 s0 = tile[BUF][(t * NQ + q) * 2 + 0][lane];
          const u64x2 s1 = tile[BUF][(t * NQ + q) * 2 + 1][lane];
          sb.s[0][0] = s0.x;
          sb.s[0][1] = s0.y;
          sb.s[1][0] = s1.x;
          sb.s[1][1] = s1.y;
#pragma unroll
          for (int r = 0; r < ROWS_W; ++r) {
            sb.d[r][0] = dbuf[(BUF * TCOLS + t) % RING][r].x;
            sb.d[r][1] = dbuf[(BUF * TCOLS + t) % RING][r].y;
          }
          if constexpr (std::is_same<ACC, AccLimb>::value)
            scan_mac_presplit<ROWS_W>(acc[q], sb);
          else
            scan_mac<ROWS_W, 2>(acc[q], sb);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    since += TCOLS;
    if (since + TCOLS > lazy) {  // fold before the lazy sums could overflow
      since = 0;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int r = 0; r < ROWS_W; ++r)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            acc[q][r][p][0].set(acc[q][r][p][0].fold(m));
            acc[q][r][p][1].set(acc[q][r][p][1].fold(m));
          }
    }
    __builtin_amdgcn_sched_barrier(0);
    stage_store(BUF ^ 1);
    __syncthreads();
  };

  stage_load(0);
#pragma unroll
  for (int t = 0; t < DIST; ++t) {
    const uint32_t col = (uint32_t)t < last_col ? (uint32_t)t : last_col;
#pragma unroll
    for (int r = 0; r < ROWS_W; ++r)
      dbuf[t][r] = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(rp[r] + (size_t)col * kN));
  }
  stage_store(0);
  __syncthreads();
  for (uint32_t tl = 0; tl < ntiles; tl += 2) {
    step(std::integral_constant<int, 0>{}, tl);
    if (tl + 1 < ntiles) step(std::integral_constant<int, 1>{}, tl + 1);
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int r = 0; r < ROWS_W; ++r) {
      const uint32_t row = row0 + r;
      if (row < rows) {
        uint64_t* o = a.out[q] + (size_t)row * 2 * kN + c0;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          o[(size_t)p * kN + 0] = acc[q][r][p][0].fold(m);
          o[(size_t)p * kN + 1] = acc[q][r][p][1].fold(m);
        }
      }
    }
}

// out[x] = sum_s part[s][x] mod q_j over ciphertext words (split reduce and
// multi-GPU fix-up share this kernel: nsplit == 1 is a pure x mod q_j).
__global__ void reduce_splits_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ part_all,
                                     uint32_t nsplit, uint64_t words, uint64_t* __restrict__ out_all,
                                     uint64_t part_qstride, uint64_t out_qstride) {
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= words) return;
  const uint64_t* part = part_all + (size_t)blockIdx.y * part_qstride;   // blockIdx.y = query of the group
  uint64_t* out = out_all + (size_t)blockIdx.y * out_qstride;
  const uint32_t j = (uint32_t)((gid >> P->logN) % P->k);
  const ModConst m = P->mod[j];
  uint64_t acc = 0;
  uint32_t s = 0;
  for (; s + 8 <= nsplit; s += 8) {  // eight partial sums in flight at a time (the launch is a few workgroups: latency-bound)
    uint64_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(s + u) * words + gid];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = add_mod(acc, reduce64(v[u], m), m.q);
  }
  for (; s < nsplit; ++s) acc = add_mod(acc, reduce64(part[(size_t)s * words + gid], m), m.q);
  out[gid] = acc;
}

// ------------------------------------------------------------------ launchers

#define PIRGPU_LAUNCH_CHECK()            \
  do {                                   \
    hipError_t e_ = hipGetLastError();   \
    if (e_ != hipSuccess) return e_;     \
  } while (0)

static inline uint32_t log2u(uint32_t N) {
  uint32_t l = 0;
  while ((1u << l) < N) ++l;
  return l;
}

// kernels with an NTT inside are compiled per ring degree and per width of the packed intermediates (ntt_kernels.hip:
// -DPIRGPU_LOGN, -DPIRGPU_PACK_BYTES; pir_amd/build.py)
#define PIRGPU_DECL_OPS(L) const NttOps* ntt_ops_##L(); const NttOps* ntt_ops_##L##_p6(); const NttOps* ntt_ops_##L##_p7();
PIRGPU_DECL_OPS(11) PIRGPU_DECL_OPS(12) PIRGPU_DECL_OPS(13) PIRGPU_DECL_OPS(14)
#undef PIRGPU_DECL_OPS

const NttOps* ntt_ops_for(uint32_t N, int pack_bytes) {
#define PIRGPU_PICK(L) case L: return pack_bytes == 6 ? ntt_ops_##L##_p6() : (pack_bytes == 7 ? ntt_ops_##L##_p7() : ntt_ops_##L());
  switch (log2u(N)) {
    PIRGPU_PICK(11) PIRGPU_PICK(12) PIRGPU_PICK(13) PIRGPU_PICK(14)
    default: return nullptr;
  }
#undef PIRGPU_PICK
}

hipError_t launch_ntt_reorder(hipStream_t st, uint32_t N, const uint64_t* in, uint64_t* out, uint64_t n_polys,
                              bool to_device, bool as_f64) {
  if (!n_polys) return hipSuccess;
  const uint64_t total = n_polys * N;
  hipLaunchKernelGGL(ntt_reorder_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, in, out, log2u(N),
                     n_polys, to_device ? 1 : 0, as_f64 ? 1 : 0);
  PIRGPU_LAUNCH_CHECK();
  return hipSuccess;
}

// Residues below 2^40 as 5 bytes for the multi-GPU exchange of row selectors: 4 words -> 5 dwords (the four low halves,
// then the four high bytes in one dword), so every access is an aligned dword and the form does not depend on how the
// buffer is cut into per-rank pieces (every piece is a multiple of 4 words).
__global__ void pack40x4_kernel(const uint64_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t quads) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= quads) return;
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const u64x2 a = reinterpret_cast<const u64x2*>(in)[2 * i], b = reinterpret_cast<const u64x2*>(in)[2 * i + 1];
  uint32_t* o = out + 5 * i;
  o[0] = (uint32_t)a[0];
  o[1] = (uint32_t)a[1];
  o[2] = (uint32_t)b[0];
  o[3] = (uint32_t)b[1];
  o[4] = (uint32_t)((a[0] >> 32) & 0xFF) | (uint32_t)((a[1] >> 32) & 0xFF) << 8 | (uint32_t)((b[0] >> 32) & 0xFF) << 16 |
         (uint32_t)((b[1] >> 32) & 0xFF) << 24;
}

__global__ void unpack40x4_kernel(const uint32_t* __restrict__ in, uint64_t* __restrict__ out, uint64_t quads) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= quads) return;
  const uint32_t* p = in + 5 * i;
  const uint32_t hi = p[4];
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  reinterpret_cast<u64x2*>(out)[2 * i] = u64x2{p[0] | (uint64_t)(hi & 0xFF) << 32, p[1] | (uint64_t)((hi >> 8) & 0xFF) << 32};
  reinterpret_cast<u64x2*>(out)[2 * i + 1] =
      u64x2{p[2] | (uint64_t)((hi >> 16) & 0xFF) << 32, p[3] | (uint64_t)(hi >> 24) << 32};
}

hipError_t launch_pack40x4(hipStream_t st, const uint64_t* in, uint32_t* out, uint64_t words) {
  const uint64_t quads = words / 4;
  hipLaunchKernelGGL(pack40x4_kernel, dim3((uint32_t)((quads + 255) / 256)), dim3(256), 0, st, in, out, quads);
  return hipGetLastError();
}

hipError_t launch_unpack40x4(hipStream_t st, const uint32_t* in, uint64_t* out, uint64_t words) {
  const uint64_t quads = words / 4;
  hipLaunchKernelGGL(unpack40x4_kernel, dim3((uint32_t)((quads + 255) / 256)), dim3(256), 0, st, in, out, quads);
  return hipGetLastError();
}

hipError_t launch_ks_combine(hipStream_t st, const DevParams* P, int mode, uint32_t N, uint32_t k,
                             const uint64_t* res_in, const uint64_t* prod, uint32_t galois_inv, uint32_t nodes,
                             uint32_t shift_pow, bool expand_step, uint32_t hi_limit, bool pack40, uint64_t* res_out,
                             int pack_bytes) {
  const uint64_t total = (uint64_t)nodes * k * N;
  const dim3 grid((uint32_t)((total + 255) / 256)), block(256);
  if (mode != kNttInt) {  // fp64 flavours: tree and products are doubles / offset-packed signed representatives
    const double* in = reinterpret_cast<const double*>(res_in);
    double* out = reinterpret_cast<double*>(res_out);
#define PIRGPU_KSC(PB_)                                                                                           \
  hipLaunchKernelGGL(ks_combine_f64_kernel<PB_>, grid, block, 0, st, P, in, prod, galois_inv, nodes, shift_pow,   \
                     expand_step ? 1 : 0, hi_limit, out)
    if (!pack40) PIRGPU_KSC(0);
    else if (pack_bytes == 5) PIRGPU_KSC(5);
    else if (pack_bytes == 6) PIRGPU_KSC(6);
    else if (pack_bytes == 7) PIRGPU_KSC(7);
    else return hipErrorInvalidValue;
#undef PIRGPU_KSC
  } else if (pack40) {
    hipLaunchKernelGGL(ks_combine_kernel<true>, grid, block, 0, st, P, res_in, prod, galois_inv, nodes, shift_pow,
                       expand_step ? 1 : 0, hi_limit, res_out);
  } else {
    hipLaunchKernelGGL(ks_combine_kernel<false>, grid, block, 0, st, P, res_in, prod, galois_inv, nodes, shift_pow,
                       expand_step ? 1 : 0, hi_limit, res_out);
  }
  PIRGPU_LAUNCH_CHECK();
  return hipSuccess;
}

// canonical u64 ciphertext words <-> the expansion tree's element type (doubles in the fp64 flavours; a plain
// copy for the integer flavour)
hipError_t launch_tree_convert(hipStream_t st, const DevParams* P, int mode, const uint64_t* in, uint64_t* out,
                               uint64_t words, bool to_tree, uint64_t chunk_words, uint64_t in_stride) {
  if (!words) return hipSuccess;
  if (!chunk_words) chunk_words = in_stride = words;
  if (mode == kNttInt) {
    if (in_stride == chunk_words) return hipMemcpyAsync(out, in, words * 8, hipMemcpyDeviceToDevice, st);
    return hipMemcpy2DAsync(out, chunk_words * 8, in, in_stride * 8, chunk_words * 8, words / chunk_words,
                            hipMemcpyDeviceToDevice, st);
  }
  const dim3 grid((uint32_t)((words + 255) / 256)), block(256);
  if (to_tree)
    hipLaunchKernelGGL(tree_import_kernel, grid, block, 0, st, P, in, reinterpret_cast<double*>(out), words, chunk_words,
                       in_stride);
  else
    hipLaunchKernelGGL(tree_export_kernel, grid, block, 0, st, P, reinterpret_cast<const double*>(in), out, words);
  PIRGPU_LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_upper_mac(hipStream_t st, const DevParams* P, const uint64_t* scratch, const MfmaPtrs& svq,
                            uint64_t* acc, uint64_t* out, uint32_t n_queries, uint32_t n_rows, uint32_t C,
                            uint32_t enc_count, uint32_t k, uint32_t N, uint32_t sv_first, uint32_t b0, uint32_t blk,
                            uint32_t n_dim, bool first, bool last, uint64_t acc_qstride, uint64_t out_qstride) {
  // Encode chunks per thread: the largest divisor of E up to 12 (E = 2 * ExpansionRatio: 8 at cfg 3, 24 at cfg 5)
  uint32_t eg = 1;
  for (uint32_t c : {12u, 8u, 6u, 4u, 3u, 2u})
    if (enc_count % c == 0) {
      eg = c;
      break;
    }
  const dim3 grid((N + 255) / 256, enc_count / eg * k, n_queries * n_rows * C);
#define PIRGPU_UPPER_MAC(EG_)                                                                                          \
  hipLaunchKernelGGL(upper_mac_kernel<EG_>, grid, dim3(256), 0, st, P, reinterpret_cast<const double*>(scratch), svq, \
                     reinterpret_cast<double*>(acc), out, n_rows, C, enc_count, sv_first, b0, blk, n_dim, first ? 1 : 0, \
                     last ? 1 : 0, acc_qstride, out_qstride)
  switch (eg) {
    case 12: PIRGPU_UPPER_MAC(12); break;
    case 8: PIRGPU_UPPER_MAC(8); break;
    case 6: PIRGPU_UPPER_MAC(6); break;
    case 4: PIRGPU_UPPER_MAC(4); break;
    case 3: PIRGPU_UPPER_MAC(3); break;
    case 2: PIRGPU_UPPER_MAC(2); break;
    default: PIRGPU_UPPER_MAC(1); break;
  }
#undef PIRGPU_UPPER_MAC
  PIRGPU_LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_monomial_shift(hipStream_t st, const DevParams* P, uint32_t N, uint32_t k, const uint64_t* in,
                                 uint32_t shift, uint64_t count, uint64_t* out) {
  uint64_t total = count * 2 * k * N;
  hipLaunchKernelGGL(monomial_shift_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, P, in, shift,
                     count, out);
  PIRGPU_LAUNCH_CHECK();
  return hipSuccess;
}

template <int ROWS, int VEC, typename ACC>
static hipError_t launch_scan_variant(hipStream_t st, const DevParams* P, uint32_t kN, uint32_t block,
                                      const uint64_t* db, const uint64_t* sv, uint64_t* out, uint32_t rows,
                                      uint32_t cols, uint64_t num_pt, uint32_t nsplit, uint32_t cols_per_split) {
  dim3 grid((kN / VEC + block - 1) / block, (rows + ROWS - 1) / ROWS, nsplit);
  hipLaunchKernelGGL((scan_kernel<ROWS, VEC, ACC>), grid, dim3(block), 0, st, P, db, sv, out, rows, cols, num_pt,
                     cols_per_split);
  return hipGetLastError();
}

hipError_t launch_scan(hipStream_t st, const DevParams* P, uint32_t N, uint32_t k, const uint64_t* db,
                       const uint64_t* sv, uint64_t* out, uint32_t rows, uint32_t cols, uint64_t num_pt,
                       uint32_t nsplit, uint32_t cols_per_split, uint32_t rows_per_thread, uint32_t block,
                       bool limb) {
  const uint32_t kN = k * N;
#define PIRGPU_SCAN_CASE(R)                                                                                        \
  case R:                                                                                                          \
    return limb ? launch_scan_variant<R, 2, AccLimb>(st, P, kN, block, db, sv, out, rows, cols, num_pt, nsplit,     \
                                                     cols_per_split)                                               \
                : launch_scan_variant<R, 2, AccWide>(st, P, kN, block, db, sv, out, rows, cols, num_pt, nsplit,     \
                                                     cols_per_split)
  switch (rows_per_thread) {
    PIRGPU_SCAN_CASE(2);
    PIRGPU_SCAN_CASE(4);
    default: return hipErrorInvalidValue;
  }
#undef PIRGPU_SCAN_CASE
}

template <int ROWS_W, int NQ, int TCOLS, typename ACC, int MINWAVES = 1>
static hipError_t launch_scan_mq_variant(hipStream_t st, const DevParams* P, uint32_t kN, const uint64_t* db,
                                         const MqArgs& a, uint32_t rows, uint32_t cols) {
  const size_t lds = (size_t)2 * TCOLS * NQ * 2 * 1024;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void*)scan_mq_kernel<ROWS_W, NQ, TCOLS, ACC, MINWAVES>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    configured = true;
  }
  dim3 grid((kN / 128) * ((rows + 4 * ROWS_W - 1) / (4 * ROWS_W)));
  hipLaunchKernelGGL((scan_mq_kernel<ROWS_W, NQ, TCOLS, ACC, MINWAVES>), grid, dim3(256), lds, st, P, db, a, rows,
                     cols);
  return hipGetLastError();
}

// Multi-query scan: nq in {1,2,4} queries per pass (batch mode).  sv[q] / out[q] as in MqArgs.
hipError_t launch_scan_mq(hipStream_t st, const DevParams* P, uint32_t N, uint32_t k, const uint64_t* db,
                          const uint64_t* const* sv, uint64_t* const* out, uint32_t nq, uint32_t rows,
                          uint32_t cols, uint32_t rows_per_wave, bool limb) {
  MqArgs a{};
  for (uint32_t q = 0; q < nq; ++q) {
    a.sv[q] = sv[q];
    a.out[q] = out[q];
  }
  const uint32_t kN = k * N;
#define PIRGPU_MQ(RW, NQ_, T)                                                                   \
  return limb ? launch_scan_mq_variant<RW, NQ_, T, AccLimb>(st, P, kN, db, a, rows, cols)        \
              : launch_scan_mq_variant<RW, NQ_, T, AccWide>(st, P, kN, db, a, rows, cols)
  if (nq == 1 && rows_per_wave == 4) { PIRGPU_MQ(4, 1, 4); }
  if (nq == 1 && rows_per_wave == 2) { PIRGPU_MQ(2, 1, 4); }
  if (nq == 2 && rows_per_wave == 2) { PIRGPU_MQ(2, 2, 4); }
  if (nq == 2 && rows_per_wave == 1) { PIRGPU_MQ(1, 2, 4); }
  if (nq == 4 && rows_per_wave == 1) { PIRGPU_MQ(1, 4, 4); }
  if (nq == 4 && rows_per_wave == 2) { PIRGPU_MQ(2, 4, 4); }
#undef PIRGPU_MQ
  return hipErrorInvalidValue;
}

hipError_t launch_reduce_splits(hipStream_t st, const DevParams* P, const uint64_t* part, uint32_t nsplit,
                                uint64_t words, uint64_t* out, uint32_t n_queries, uint64_t part_qstride,
                                uint64_t out_qstride) {
  if (!words || !n_queries) return hipSuccess;
  hipLaunchKernelGGL(reduce_splits_kernel, dim3((uint32_t)((words + 255) / 256), n_queries), dim3(256), 0, st, P, part,
                     nsplit, words, out, part_qstride, out_qstride);
  PIRGPU_LAUNCH_CHECK();
  return hipSuccess;
}

}  // namespace pirgpu
