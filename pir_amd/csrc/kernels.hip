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
//   reencode_lift_ntt_kernel  CiphertextReencoder::Encode + plain NTT    database.cpp:218,225-228, ct_reencoder.cpp:40-71
//   upper_mac_kernel          multiply_plain + add_inplace upper levels  database.cpp:229-230,238-247
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_params.h"
#include "kernels.h"

namespace pirgpu {

typedef unsigned __int128 u128;

// ------------------------------------------------------------------ arithmetic

__device__ __forceinline__ uint64_t mul_shoup(uint64_t x, uint64_t w, uint64_t ws, uint64_t q) {
  uint64_t h = __umul64hi(x, ws);
  uint64_t r = x * w - h * q;
  return r >= q ? r - q : r;
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

// ------------------------------------------------------------------ NTT in LDS

// Forward negacyclic NTT of the N-point polynomial held in LDS (Cooley-Tukey,
// natural order in, bit-reversed order out; twiddles psi^bitrev(m+i)).
__device__ __forceinline__ void ntt_fwd_lds(uint64_t* s, const DevParams* __restrict__ P, int mi) {
  const uint32_t N = P->N;
  const uint64_t q = P->mod[mi].q;
  const uint64_t* __restrict__ w = P->tab[mi].w;
  const uint64_t* __restrict__ ws = P->tab[mi].ws;
  const uint32_t half = N >> 1;
  uint32_t logt = P->logN;
  for (uint32_t m = 1; m < N; m <<= 1) {
    --logt;
    const uint32_t t = 1u << logt;
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < half; b += blockDim.x) {
      uint32_t i = b >> logt, j = b & (t - 1);
      uint32_t idx = (i << (logt + 1)) + j;
      uint64_t W = w[m + i], Ws = ws[m + i];
      uint64_t u = s[idx], v = mul_shoup(s[idx + t], W, Ws, q);
      s[idx] = add_mod(u, v, q);
      s[idx + t] = sub_mod(u, v, q);
    }
  }
  __syncthreads();
}

// Inverse negacyclic NTT (Gentleman-Sande, bit-reversed in, natural out, scaled by N^-1).
__device__ __forceinline__ void ntt_inv_lds(uint64_t* s, const DevParams* __restrict__ P, int mi) {
  const uint32_t N = P->N;
  const uint64_t q = P->mod[mi].q;
  const uint64_t* __restrict__ iw = P->tab[mi].iw;
  const uint64_t* __restrict__ iws = P->tab[mi].iws;
  const uint32_t half = N >> 1;
  uint32_t logt = 0;
  for (uint32_t m = N; m > 1; m >>= 1) {
    const uint32_t h = m >> 1, t = 1u << logt;
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < half; b += blockDim.x) {
      uint32_t i = b >> logt, j = b & (t - 1);
      uint32_t idx = (i << (logt + 1)) + j;
      uint64_t W = iw[h + i], Ws = iws[h + i];
      uint64_t u = s[idx], v = s[idx + t];
      s[idx] = add_mod(u, v, q);
      s[idx + t] = mul_shoup(sub_mod(u, v, q), W, Ws, q);
    }
    ++logt;
  }
  __syncthreads();
  const uint64_t ninv = P->tab[mi].ninv, ninvs = P->tab[mi].ninvs;
  for (uint32_t i = threadIdx.x; i < N; i += blockDim.x) s[i] = mul_shoup(s[i], ninv, ninvs, q);
  __syncthreads();
}

extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

// ------------------------------------------------------------------ batched NTT

// One workgroup per polynomial; modulus index = mod_base + (poly % mod_period).
__global__ void ntt_batch_kernel(const DevParams* __restrict__ P, uint64_t* __restrict__ data, uint32_t mod_period,
                                 uint32_t mod_base, int inverse) {
  uint64_t* s = reinterpret_cast<uint64_t*>(smem_raw);
  const uint32_t N = P->N;
  const int mi = mod_base + (blockIdx.x % mod_period);
  uint64_t* poly = data + (size_t)blockIdx.x * N;
  for (uint32_t i = threadIdx.x; i < N; i += blockDim.x) s[i] = poly[i];
  if (inverse)
    ntt_inv_lds(s, P, mi);
  else
    ntt_fwd_lds(s, P, mi);
  for (uint32_t i = threadIdx.x; i < N; i += blockDim.x) poly[i] = s[i];
}

// Out-of-place forward NTT of ciphertexts: src[ct][2][k][N] (coefficient form)
// -> dst[ct][2][k][N]; used to put the expanded selection vector into NTT form.
__global__ void ct_ntt_fwd_oop_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ src,
                                      uint64_t* __restrict__ dst) {
  uint64_t* s = reinterpret_cast<uint64_t*>(smem_raw);
  const uint32_t N = P->N;
  const int mi = blockIdx.x % P->k;
  const uint64_t* in = src + (size_t)blockIdx.x * N;
  uint64_t* out = dst + (size_t)blockIdx.x * N;
  for (uint32_t i = threadIdx.x; i < N; i += blockDim.x) s[i] = in[i];
  ntt_fwd_lds(s, P, mi);
  for (uint32_t i = threadIdx.x; i < N; i += blockDim.x) out[i] = s[i];
}

// ------------------------------------------------------------------ database encode

// grid = (n_pt, k).  Source is either pre-encoded coefficients (coeffs != null)
// or raw item bytes packed MSB-first into bits-wide coefficients.
__global__ void db_encode_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ coeffs,
                                 const uint8_t* __restrict__ bytes, uint64_t bytes_per_pt, uint64_t total_bytes,
                                 uint32_t bits, uint64_t* __restrict__ db) {
  uint64_t* s = reinterpret_cast<uint64_t*>(smem_raw);
  const uint32_t N = P->N, k = P->k;
  const uint32_t j = blockIdx.y;
  const uint64_t pt = blockIdx.x;
  const ModConst m = P->mod[j];
  const uint64_t thr = P->plain_thr, inc = P->lift_inc[j];
  uint64_t L = 0;
  const uint8_t* src = nullptr;
  if (!coeffs) {
    uint64_t start = pt * bytes_per_pt;
    L = start >= total_bytes ? 0 : (total_bytes - start < bytes_per_pt ? total_bytes - start : bytes_per_pt);
    src = bytes + start;
  }
  for (uint32_t c = threadIdx.x; c < N; c += blockDim.x) {
    uint64_t v;
    if (coeffs) {
      v = coeffs[pt * N + c];
    } else {
      v = 0;
      uint64_t bitpos = (uint64_t)c * bits;
      uint64_t byte = bitpos >> 3;
      uint32_t off = (uint32_t)(bitpos & 7);
      int need = (int)bits;
      while (need > 0) {
        uint32_t B = byte < L ? src[byte] : 0u;
        int avail = 8 - (int)off;
        int take = avail < need ? avail : need;
        v = (v << take) | ((B >> (avail - take)) & ((1u << take) - 1u));
        need -= take;
        off = 0;
        ++byte;
      }
    }
    uint64_t r = reduce64(v, m);
    if (v >= thr) r = add_mod(r, inc >= m.q ? inc - m.q : inc, m.q);
    s[c] = r;
  }
  ntt_fwd_lds(s, P, j);
  uint64_t* out = db + (pt * k + j) * N;
  for (uint32_t i = threadIdx.x; i < N; i += blockDim.x) out[i] = s[i];
}

// ------------------------------------------------------------------ expansion

// One level of the expansion tree, part 1: for node n and key-level modulus I,
//   S[c][I] = sum_J NTT_I(sigma_g(c1)_J mod m_I) (.) K[J][c][I]   then INTT_I.
// grid = (nodes, k+1); block = N/16 threads; prod layout [node][2][k+1][N].
__global__ void ks_main_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ res_in,
                               const uint64_t* __restrict__ key, uint32_t galois_elt, uint64_t* __restrict__ prod) {
  uint64_t* s = reinterpret_cast<uint64_t*>(smem_raw);
  const uint32_t N = P->N, k = P->k, km = k + 1, logN = P->logN;
  const uint32_t node = blockIdx.x, I = blockIdx.y;
  const ModConst mI = P->mod[I];
  const uint64_t* c1 = res_in + ((size_t)node * 2 + 1) * k * N;  // poly 1
  uint64_t acc0[kNttElemsPerThread], acc1[kNttElemsPerThread];
#pragma unroll
  for (int e = 0; e < kNttElemsPerThread; ++e) acc0[e] = acc1[e] = 0;
  for (uint32_t J = 0; J < k; ++J) {
    const uint64_t qJ = P->mod[J].q;
    const uint64_t* src = c1 + (size_t)J * N;
    __syncthreads();
    // sigma_g in coefficient form (SEAL GaloisTool::apply_galois), then reduce mod m_I
    for (uint32_t i = threadIdx.x; i < N; i += blockDim.x) {
      uint32_t raw = i * galois_elt;
      uint32_t idx = raw & (N - 1);
      uint64_t v = src[i];
      if ((raw >> logN) & 1) v = neg_mod(v, qJ);
      s[idx] = reduce64(v, mI);
    }
    ntt_fwd_lds(s, P, I);
    const uint64_t* k0 = key + (((size_t)J * 2 + 0) * km + I) * N;
    const uint64_t* k1 = key + (((size_t)J * 2 + 1) * km + I) * N;
#pragma unroll
    for (int e = 0; e < kNttElemsPerThread; ++e) {
      uint32_t pos = threadIdx.x + e * blockDim.x;
      uint64_t x = s[pos];
      acc0[e] = add_mod(acc0[e], mul_mod(x, k0[pos], mI), mI.q);
      acc1[e] = add_mod(acc1[e], mul_mod(x, k1[pos], mI), mI.q);
    }
  }
  uint64_t* out0 = prod + (((size_t)node * 2 + 0) * km + I) * N;
  uint64_t* out1 = prod + (((size_t)node * 2 + 1) * km + I) * N;
  __syncthreads();
#pragma unroll
  for (int e = 0; e < kNttElemsPerThread; ++e) s[threadIdx.x + e * blockDim.x] = acc0[e];
  ntt_inv_lds(s, P, I);
  for (uint32_t i = threadIdx.x; i < N; i += blockDim.x) out0[i] = s[i];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < kNttElemsPerThread; ++e) s[threadIdx.x + e * blockDim.x] = acc1[e];
  ntt_inv_lds(s, P, I);
  for (uint32_t i = threadIdx.x; i < N; i += blockDim.x) out1[i] = s[i];
}

// Part 2: divide-and-round by the special prime, add sigma_g(c0), then the tree
// butterfly   res_out[n] = a + g,   res_out[n + nodes] = x^(-2^j) * (a - g)
// (identical residues to the reference's two negacyclic shifts + adds, since
// x^-(N+2^j) = -x^(-2^j)).  With expand_step == 0 only g is written (plain
// substitute_power_x_inplace).  One thread per (node, residue, coefficient).
__global__ void ks_combine_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ res_in,
                                  const uint64_t* __restrict__ prod, uint32_t galois_inv, uint32_t nodes,
                                  uint32_t shift_pow, int expand_step, uint64_t* __restrict__ res_out) {
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
    const uint64_t* pr = prod + ((size_t)node * 2 + comp) * km * N;
    uint64_t r = add_mod(pr[(size_t)k * N + i], P->p_half, mp.q);
    uint64_t delta = sub_mod(reduce64(r, mj), P->p_half_mod[j], q);
    uint64_t v = sub_mod(pr[(size_t)j * N + i], delta, q);
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
#pragma unroll
  for (int comp = 0; comp < 2; ++comp) {
    size_t off = ((size_t)comp * k + j) * N;
    uint64_t a = a_ct[off + i];
    lo[off + i] = add_mod(a, g[comp], q);
    uint64_t d = sub_mod(a, g[comp], q);
    if (sneg) d = neg_mod(d, q);
    hi[off + sidx] = d;
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

typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));

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

// Base case of DatabaseMultiplier::multiply fused over a whole row:
//   out[split][row][p][j][c] = sum_{col in split} sv[col][p][j][c] * db[row*cols + col][j][c]  mod q_j
// Each thread owns VEC adjacent coefficients of one residue for ROWS rows and
// streams the database exactly once (16-byte loads, fully coalesced); products
// accumulate lazily in 128 bits and are reduced once (or every lazy_limit terms).
// grid = (k*N / (VEC*block), ceil(rows / ROWS), nsplit).
template <int ROWS, int VEC>
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
  const uint32_t lazy = P->lazy_limit;

  u128 acc[ROWS][2][VEC];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[r][p][v] = 0;

  // number of valid columns per row (database may end inside the last row)
  uint32_t ncol[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    uint32_t row = row0 + r;
    uint64_t first = (uint64_t)row * cols;
    uint64_t avail = (row < rows && first < num_pt) ? num_pt - first : 0;
    ncol[r] = avail > cols ? cols : (uint32_t)avail;
  }

  uint32_t since = 0;
  for (uint32_t col = col_begin; col < col_end; ++col) {
    const uint64_t* svp = sv + (size_t)col * 2 * kN + c0;
    uint64_t s0[VEC], s1[VEC];
    load_vec<VEC>(svp, s0);
    load_vec<VEC>(svp + kN, s1);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      if (col < ncol[r]) {
        const uint64_t* dp = db + ((size_t)(row0 + r) * cols + col) * kN + c0;
        uint64_t d[VEC];
        load_vec<VEC>(dp, d);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          acc[r][0][v] += (u128)s0[v] * d[v];
          acc[r][1][v] += (u128)s1[v] * d[v];
        }
      }
    }
    if (++since == lazy) {
      since = 0;
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int v = 0; v < VEC; ++v)
            acc[r][p][v] = reduce128((uint64_t)acc[r][p][v], (uint64_t)(acc[r][p][v] >> 64), m);
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
        for (int v = 0; v < VEC; ++v)
          o[(size_t)p * kN + v] = reduce128((uint64_t)acc[r][p][v], (uint64_t)(acc[r][p][v] >> 64), m);
    }
  }
}

// out[x] = sum_s part[s][x] mod q_j over ciphertext words (split reduce and
// multi-GPU fix-up share this kernel: nsplit == 1 is a pure x mod q_j).
__global__ void reduce_splits_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ part,
                                     uint32_t nsplit, uint64_t words, uint64_t* __restrict__ out) {
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= words) return;
  const uint32_t j = (uint32_t)((gid >> P->logN) % P->k);
  const ModConst m = P->mod[j];
  uint64_t acc = 0;
  for (uint32_t s = 0; s < nsplit; ++s) acc = add_mod(acc, reduce64(part[(size_t)s * words + gid], m), m.q);
  out[gid] = acc;
}

// ------------------------------------------------------------------ upper levels

// CiphertextReencoder::Encode chunk e of source ciphertext c, lifted to residue
// jt and forward-NTT'd: pt[c][e][jt][N].  grid = (n_src, enc_count, k).
__global__ void reencode_lift_ntt_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ src,
                                         uint64_t* __restrict__ pt) {
  uint64_t* s = reinterpret_cast<uint64_t*>(smem_raw);
  const uint32_t N = P->N, k = P->k;
  const uint32_t c = blockIdx.x, e = blockIdx.y, jt = blockIdx.z;
  const ModConst m = P->mod[jt];
  const uint32_t sp = P->enc_poly[e], sj = P->enc_res[e], sh = P->enc_shift[e];
  const uint64_t mask = (1ull << P->enc_bits) - 1;
  const uint64_t thr = P->plain_thr;
  const uint64_t inc = P->lift_inc[jt] >= m.q ? P->lift_inc[jt] - m.q : P->lift_inc[jt];
  const uint64_t* in = src + (((size_t)c * 2 + sp) * k + sj) * N;
  for (uint32_t i = threadIdx.x; i < N; i += blockDim.x) {
    uint64_t v = (in[i] >> sh) & mask;
    uint64_t r = reduce64(v, m);
    if (v >= thr) r = add_mod(r, inc, m.q);
    s[i] = r;
  }
  ntt_fwd_lds(s, P, jt);
  uint64_t* out = pt + (((size_t)c * P->enc_count + e) * k + jt) * N;
  for (uint32_t i = threadIdx.x; i < N; i += blockDim.x) out[i] = s[i];
}

// Upper-level accumulate:  out[r][cc * E + e][p][j][i] =
//    sum_{ii < nchild(r)} sv[ii][p][j][i] * pt[(child0(r) + ii) * C + cc][e][j][i]   mod q_j
// where C = ciphertexts per child and E = enc_count.  One thread per output word.
__global__ void upper_mac_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ sv,
                                 const uint64_t* __restrict__ pt, uint32_t n_rows, uint32_t n_dim,
                                 uint32_t n_children_total, uint32_t sv_first, uint32_t C,
                                 uint64_t* __restrict__ out) {
  const uint32_t N = P->N, k = P->k, E = P->enc_count;
  const uint64_t words_per_row = (uint64_t)C * E * 2 * k * N;
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= words_per_row * n_rows) return;
  const uint32_t i = (uint32_t)(gid & (N - 1));
  uint64_t rest = gid >> P->logN;
  const uint32_t j = (uint32_t)(rest % k);
  rest /= k;
  const uint32_t p = (uint32_t)(rest & 1);
  rest >>= 1;
  const uint32_t e = (uint32_t)(rest % E);
  rest /= E;
  const uint32_t cc = (uint32_t)(rest % C);
  const uint32_t r = (uint32_t)(rest / C);
  const ModConst m = P->mod[j];
  const uint32_t child0 = r * n_dim;
  uint32_t nchild = n_children_total > child0 ? n_children_total - child0 : 0;
  if (nchild > n_dim) nchild = n_dim;
  const uint32_t lazy = P->lazy_limit;
  u128 acc = 0;
  uint32_t since = 0;
  for (uint32_t ii = 0; ii < nchild; ++ii) {
    uint64_t a = sv[(((size_t)(sv_first + ii) * 2 + p) * k + j) * N + i];
    uint64_t b = pt[((((size_t)(child0 + ii) * C + cc) * E + e) * k + j) * N + i];
    acc += (u128)a * b;
    if (++since == lazy) {
      since = 0;
      acc = reduce128((uint64_t)acc, (uint64_t)(acc >> 64), m);
    }
  }
  out[gid] = reduce128((uint64_t)acc, (uint64_t)(acc >> 64), m);
}

// ------------------------------------------------------------------ launchers

static inline uint32_t ntt_threads(uint32_t N) { return N / kNttElemsPerThread; }
static inline size_t ntt_lds(uint32_t N) { return (size_t)N * sizeof(uint64_t); }

#define PIRGPU_LAUNCH_CHECK()            \
  do {                                   \
    hipError_t e_ = hipGetLastError();   \
    if (e_ != hipSuccess) return e_;     \
  } while (0)

hipError_t configure_kernels(uint32_t N) {
  // dynamic LDS beyond 64 KiB (N = 16384 -> 128 KiB of the CU's 160 KiB)
  int bytes = (int)ntt_lds(N);
  hipError_t e;
  if ((e = hipFuncSetAttribute((const void*)ntt_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)))
    return e;
  if ((e = hipFuncSetAttribute((const void*)ct_ntt_fwd_oop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               bytes)))
    return e;
  if ((e = hipFuncSetAttribute((const void*)db_encode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)))
    return e;
  if ((e = hipFuncSetAttribute((const void*)ks_main_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)))
    return e;
  if ((e = hipFuncSetAttribute((const void*)reencode_lift_ntt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               bytes)))
    return e;
  return hipSuccess;
}

hipError_t launch_ntt_batch(hipStream_t st, const DevParams* P, uint32_t N, uint64_t* data, uint64_t n_polys,
                            uint32_t mod_period, uint32_t mod_base, bool inverse) {
  if (!n_polys) return hipSuccess;
  hipLaunchKernelGGL(ntt_batch_kernel, dim3((uint32_t)n_polys), dim3(ntt_threads(N)), ntt_lds(N), st, P, data,
                     mod_period, mod_base, inverse ? 1 : 0);
  PIRGPU_LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_ct_ntt_fwd_oop(hipStream_t st, const DevParams* P, uint32_t N, uint32_t k, const uint64_t* src,
                                 uint64_t* dst, uint64_t n_cts) {
  if (!n_cts) return hipSuccess;
  hipLaunchKernelGGL(ct_ntt_fwd_oop_kernel, dim3((uint32_t)(n_cts * 2 * k)), dim3(ntt_threads(N)), ntt_lds(N), st,
                     P, src, dst);
  PIRGPU_LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_db_encode(hipStream_t st, const DevParams* P, uint32_t N, uint32_t k, const uint64_t* coeffs,
                            const uint8_t* bytes, uint64_t bytes_per_pt, uint64_t total_bytes, uint32_t bits,
                            uint64_t n_pt, uint64_t* db) {
  if (!n_pt) return hipSuccess;
  hipLaunchKernelGGL(db_encode_kernel, dim3((uint32_t)n_pt, k), dim3(ntt_threads(N)), ntt_lds(N), st, P, coeffs,
                     bytes, bytes_per_pt, total_bytes, bits, db);
  PIRGPU_LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_ks_level(hipStream_t st, const DevParams* P, uint32_t N, uint32_t k, const uint64_t* res_in,
                           const uint64_t* key, uint32_t galois_elt, uint32_t galois_inv, uint32_t nodes,
                           uint32_t shift_pow, bool expand_step, uint64_t* prod, uint64_t* res_out) {
  hipLaunchKernelGGL(ks_main_kernel, dim3(nodes, k + 1), dim3(ntt_threads(N)), ntt_lds(N), st, P, res_in, key,
                     galois_elt, prod);
  PIRGPU_LAUNCH_CHECK();
  uint64_t total = (uint64_t)nodes * k * N;
  hipLaunchKernelGGL(ks_combine_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, P, res_in, prod,
                     galois_inv, nodes, shift_pow, expand_step ? 1 : 0, res_out);
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

hipError_t launch_scan(hipStream_t st, const DevParams* P, uint32_t N, uint32_t k, const uint64_t* db,
                       const uint64_t* sv, uint64_t* out, uint32_t rows, uint32_t cols, uint64_t num_pt,
                       uint32_t nsplit, uint32_t cols_per_split) {
  constexpr int ROWS = 4, VEC = 2;
  const uint32_t kN = k * N;
  const uint32_t block = 256;
  dim3 grid((kN / VEC + block - 1) / block, (rows + ROWS - 1) / ROWS, nsplit);
  hipLaunchKernelGGL((scan_kernel<ROWS, VEC>), grid, dim3(block), 0, st, P, db, sv, out, rows, cols, num_pt,
                     cols_per_split);
  PIRGPU_LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_reduce_splits(hipStream_t st, const DevParams* P, const uint64_t* part, uint32_t nsplit,
                                uint64_t words, uint64_t* out) {
  if (!words) return hipSuccess;
  hipLaunchKernelGGL(reduce_splits_kernel, dim3((uint32_t)((words + 255) / 256)), dim3(256), 0, st, P, part, nsplit,
                     words, out);
  PIRGPU_LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_reencode_lift_ntt(hipStream_t st, const DevParams* P, uint32_t N, uint32_t k, uint32_t enc_count,
                                    const uint64_t* src, uint64_t n_src, uint64_t* pt) {
  if (!n_src) return hipSuccess;
  hipLaunchKernelGGL(reencode_lift_ntt_kernel, dim3((uint32_t)n_src, enc_count, k), dim3(ntt_threads(N)),
                     ntt_lds(N), st, P, src, pt);
  PIRGPU_LAUNCH_CHECK();
  return hipSuccess;
}

hipError_t launch_upper_mac(hipStream_t st, const DevParams* P, uint32_t N, uint32_t k, uint32_t enc_count,
                            const uint64_t* sv, const uint64_t* pt, uint32_t n_rows, uint32_t n_dim,
                            uint32_t n_children_total, uint32_t sv_first, uint32_t C, uint64_t* out) {
  uint64_t total = (uint64_t)n_rows * C * enc_count * 2 * k * N;
  if (!total) return hipSuccess;
  hipLaunchKernelGGL(upper_mac_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, P, sv, pt, n_rows,
                     n_dim, n_children_total, sv_first, C, out);
  PIRGPU_LAUNCH_CHECK();
  return hipSuccess;
}

}  // namespace pirgpu
