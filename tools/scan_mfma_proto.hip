// Prototype of the digit-sliced MFMA database scan (standalone: synthetic data, reference check, timing).
//   out[q][r][comp][j] = sum_c DB[r][c][j] * S[q][c][comp][j]  mod q_j
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/scan_mfma_proto.hip -o tools/scan_mfma_proto
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
#ifndef SWZ
#define SWZ 0
#endif
#ifndef NTS
#define NTS 0
#endif
#ifndef LAYOUT2
#define LAYOUT2 0
#endif
// LAYOUT2: database tiles stored [j][rt][ks][a][g][r16][16B] (a whole A tile = 1 KB contiguous, K padded to 64)
#define DBOFF(rt, kg, a) (LAYOUT2 ? ((((size_t)(rt) * KSTEPS + ((kg) >> 2)) * L + (a)) * 1024 + ((kg) & 3) * 256) : (((size_t)(rt) * KG + (kg)) * L + (a)) * 256)
#ifndef NO_MFMA
#define NO_MFMA 0
#endif
#ifndef NO_STORE
#define NO_STORE 0
#endif
#ifndef NO_RECOMB
#define NO_RECOMB 0
#endif
typedef unsigned __int128 u128;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

struct Mod { uint64_t q, bias, mu; };   // bias = q * 2^s >= 2^58.., c40[g] = 2^(40 g) mod q

__device__ __forceinline__ uint64_t hash64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}

// edge != 0: residues drawn from the boundary cases of the centring / digit decomposition instead of uniform
__global__ void fill_kernel(uint64_t* p, size_t n, uint64_t q0, uint64_t q1, uint32_t N, uint32_t kN, uint64_t seed,
                            int edge) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t j = (uint32_t)(i % kN);
  uint64_t q = j < N ? q0 : q1;
  uint64_t h = hash64(i * 0x9E3779B97F4A7C15ULL + seed);
  if (!edge) { p[i] = h % q; return; }
  const uint64_t half = q >> 1;
  const uint64_t cases[16] = {0, 1, q - 1, q - 2, half, half + 1, half - 1, half + 2,
                              0x7F, 0x80, 0x81, 0x7F7F7F7F7FULL % q, 0x8080808080ULL % q, 0x807F807F80ULL % q,
                              q - 0x80, q - 0x8080};
  uint64_t v = cases[h & 15];
  if ((h >> 4) & 1) v = (v + (((h >> 8) & 3) << 8)) % q;   // perturb a middle digit
  p[i] = v;
}

// reference: one thread per (q, r, comp, j)
__global__ void ref_kernel(const uint64_t* db, const uint64_t* sel, uint64_t* out, uint32_t R, uint32_t C, uint32_t kN,
                           uint32_t N, uint32_t NQ, uint64_t q0, uint64_t q1) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)NQ * R * 2 * kN;
  if (i >= total) return;
  uint32_t j = (uint32_t)(i % kN);
  uint32_t comp = (uint32_t)((i / kN) % 2);
  uint32_t r = (uint32_t)((i / kN / 2) % R);
  uint32_t q = (uint32_t)(i / kN / 2 / R);
  uint64_t m = j < N ? q0 : q1;
  u128 acc = 0;
  for (uint32_t c = 0; c < C; ++c)
    acc += (u128)db[((size_t)r * C + c) * kN + j] * sel[(((size_t)q * C + c) * 2 + comp) * kN + j];
  out[i] = (uint64_t)(acc % m);
}

// ---------------------------------------------------------------- packing

template <int L>
__device__ __forceinline__ void digits(uint64_t x, uint64_t q, int8_t (&d)[L]) {
  int64_t v = x > (q >> 1) ? (int64_t)x - (int64_t)q : (int64_t)x;
#pragma unroll
  for (int a = 0; a < L; ++a) {
    d[a] = (int8_t)(v & 0xFF);
    v = (v - d[a]) >> 8;
  }
}

// DB u64 [R][C][kN] -> packed [j][rt][kg][a][r16][c16] bytes.  block = 256 threads: 16 r x 16 j; grid = (kN/16, RT, KG)
template <int L>
__global__ void __launch_bounds__(256)
db_pack_kernel(const uint64_t* __restrict__ db, uint8_t* __restrict__ dbp, uint32_t R, uint32_t C, uint32_t kN,
               uint32_t N, uint32_t RT, uint32_t KG, uint64_t q0, uint64_t q1) {
  const uint32_t j = blockIdx.x * 16 + (threadIdx.x & 15);
  const uint32_t r16 = threadIdx.x >> 4;
  const uint32_t rt = blockIdx.y, kg = blockIdx.z;
  const uint32_t r = rt * 16 + r16;
  const uint64_t q = j < N ? q0 : q1;
  uint8_t o[L][16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const uint32_t c = kg * 16 + t;
    uint64_t x = 0;
    if (r < R && c < C) x = db[((size_t)r * C + c) * kN + j];
    int8_t d[L];
    digits<L>(x, q, d);
#pragma unroll
    for (int a = 0; a < L; ++a) o[a][t] = (uint8_t)d[a];
  }
#pragma unroll
  for (int a = 0; a < L; ++a) {
    uint4 v;
    __builtin_memcpy(&v, o[a], 16);
    const uint32_t KSTEPS = (KG + 3) / 4;
    const size_t slab = LAYOUT2 ? (size_t)RT * KSTEPS * L * 1024 : (size_t)RT * KG * L * 256;
    *reinterpret_cast<uint4*>(dbp + (size_t)j * slab + DBOFF(rt, kg, a) + r16 * 16) = v;
  }
}

// selectors u64 [NQ][C][2][kN] -> packed [j][kg][b][x][c16]; block 256 = 16 x * 16 j; grid = (kN/16, KG)
template <int L>
__global__ void __launch_bounds__(256)
sel_pack_kernel(const uint64_t* __restrict__ sel, uint8_t* __restrict__ selp, uint32_t C, uint32_t kN, uint32_t N,
                uint32_t KG, uint32_t NQ, uint64_t q0, uint64_t q1) {
  const uint32_t j = blockIdx.x * 16 + (threadIdx.x & 15);
  const uint32_t x = threadIdx.x >> 4;
  const uint32_t kg = blockIdx.y;
  const uint64_t q = j < N ? q0 : q1;
  const uint32_t qi = x >> 1, comp = x & 1;
  uint8_t o[L][16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const uint32_t c = kg * 16 + t;
    uint64_t v = 0;
    if (qi < NQ && c < C) v = sel[(((size_t)qi * C + c) * 2 + comp) * kN + j];
    int8_t d[L];
    digits<L>(v, q, d);
#pragma unroll
    for (int b = 0; b < L; ++b) o[b][t] = (uint8_t)d[b];
  }
#pragma unroll
  for (int b = 0; b < L; ++b) {
    uint4 v;
    __builtin_memcpy(&v, o[b], 16);
    *reinterpret_cast<uint4*>(selp + (((size_t)j * KG + kg) * L + b) * 256 + x * 16) = v;
  }
}

// ---------------------------------------------------------------- the scan

__device__ __forceinline__ v4i load16(const uint8_t* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const v4i*>(p));
}

// x < 2^63 -> x mod q with mu = floor(2^64 / q)
__device__ __forceinline__ uint64_t red64(uint64_t x, const Mod& m) {
  const uint64_t qh = __umul64hi(x, m.mu);
  uint64_t r = x - qh * m.q;
  if (r >= m.q) r -= m.q;
  if (r >= m.q) r -= m.q;
  return r;
}

template <int L, int KS>
__global__ void __launch_bounds__(512)
scan_mfma_kernel(const uint8_t* __restrict__ dbp, const uint8_t* __restrict__ selp, uint64_t* __restrict__ out,
                 uint32_t RT, uint32_t KG, uint32_t R, uint32_t NQ, uint32_t kN, uint32_t N, Mod m0, Mod m1) {
  __shared__ uint64_t stage[2][16][16][8];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int g = l >> 4, i16 = l & 15;
#if SWZ
  const uint32_t nb = gridDim.x;                      // consecutive j-blocks on the same XCD (block b -> XCD b % 8)
  const uint32_t j0 = ((blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3)) * 8;
#else
  const uint32_t j0 = blockIdx.x * 8;
#endif
  const uint32_t j = j0 + w;
  const Mod m = j < N ? m0 : m1;

  v4i B[KS][L];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const uint32_t kg = ks * 4 + g;
#pragma unroll
    for (int b = 0; b < L; ++b) {
      B[ks][b] = v4i{0, 0, 0, 0};
      if (kg < KG) B[ks][b] = *reinterpret_cast<const v4i*>(selp + (((size_t)j * KG + kg) * L + b) * 256 + i16 * 16);
    }
  }
  const uint32_t KSTEPS = (KG + 3) / 4;
  const uint8_t* abase = dbp + (size_t)j * (LAYOUT2 ? (size_t)RT * KSTEPS * L * 1024 : (size_t)RT * KG * L * 256) + i16 * 16;
  const uint32_t nx = 2 * NQ;

  v4i A[KS][L];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const uint32_t kg = ks * 4 + g;
#pragma unroll
    for (int a = 0; a < L; ++a) {
      A[ks][a] = v4i{0, 0, 0, 0};
      if (kg < KG) A[ks][a] = load16(abase + DBOFF(0, kg, a));
    }
  }

  for (uint32_t rt = 0; rt < RT; ++rt) {
    // diagonal accumulators: T[s] = sum_{a+b=s} sum_k A_a B_b  (|T| < 5 * 2^14 * 64 KS < 2^24)
    v4i T[2 * L - 1];
#pragma unroll
    for (int s = 0; s < 2 * L - 1; ++s) T[s] = v4i{0, 0, 0, 0};
    const uint32_t rtn = rt + 1 < RT ? rt + 1 : rt;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      // order the L*L products so that consecutive MFMAs hit different accumulators
#pragma unroll
      for (int off = 0; off < L; ++off)
#pragma unroll
        for (int a = 0; a < L; ++a) {
          const int b = (a + off) % L;
          if (NO_MFMA) { if (b == 0) T[a] ^= A[ks][a]; } else
          T[a + b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[ks][a], B[ks][b], T[a + b], 0, 0, 0);
        }
      // refill this ring slot with the next row tile's step
      const uint32_t kg = ks * 4 + g;
      if (kg < KG) {
#pragma unroll
        for (int a = 0; a < L; ++a) A[ks][a] = load16(abase + DBOFF(rtn, kg, a));
      }
    }
    // value = sum_s T_s 2^(8 s) = G0 + G1 2^40, reduced mod q
    const int buf = rt & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int64_t G0 = 0, G1 = 0;
#pragma unroll
      for (int s = 0; s < 5; ++s) G0 += (int64_t)T[s][i] << (8 * s);
#pragma unroll
      for (int s = 5; s < 2 * L - 1; ++s) G1 += (int64_t)T[s][i] << (8 * (s - 5));
      uint64_t r;
      if (NO_RECOMB) {
        r = (uint64_t)(G0 ^ G1);
      } else {
        r = red64((uint64_t)(G1 + (int64_t)m.bias), m);
        r = red64(r << 20, m);
        r = red64((r << 20) + (uint64_t)(G0 + (int64_t)m.bias), m);
      }
      stage[buf][g * 4 + i][i16][w] = r;
    }
    __syncthreads();
    // cooperative store: 256 (r16, x) runs of 8 j = 64 B; 512 threads x 16 B = 2 rounds
#pragma unroll
    for (int round = 0; round < 2; ++round) {
      const int run = round * 128 + (threadIdx.x >> 2);
      const int part = threadIdx.x & 3;
      const int r16 = run >> 4, x = run & 15;
      const uint32_t r = rt * 16 + r16;
      if (x < (int)nx && r < R && !NO_STORE) {
        const v4i v = *reinterpret_cast<const v4i*>(&stage[buf][r16][x][part * 2]);
        uint64_t* dst = out + ((((size_t)(x >> 1) * R + r) * 2 + (x & 1)) * kN + j0 + part * 2);
        if (NTS) __builtin_nontemporal_store(v, reinterpret_cast<v4i*>(dst)); else *reinterpret_cast<v4i*>(dst) = v;
      }
    }
  }
}

static uint64_t powmod(uint64_t a, uint64_t e, uint64_t q) {
  u128 r = 1, b = a % q;
  while (e) { if (e & 1) r = r * b % q; b = b * b % q; e >>= 1; }
  return (uint64_t)r;
}

int main(int argc, char** argv) {
  const uint32_t N = 4096, kN = 8192;
  uint32_t R = argc > 1 ? atoi(argv[1]) : 162, C = argc > 2 ? atoi(argv[2]) : 162, NQ = argc > 3 ? atoi(argv[3]) : 8;
  const int iters = argc > 4 ? atoi(argv[4]) : 20;
  const uint64_t q0 = 0xffffee001ULL, q1 = 0xffffc4001ULL;
  constexpr int L = 5, KS = 3;
  const uint32_t RT = (R + 15) / 16, KG = (C + 15) / 16;
  if (KG > KS * 4) { printf("C too large for KS\n"); return 1; }
  Mod m[2];
  for (int i = 0; i < 2; ++i) {
    uint64_t q = i ? q1 : q0;
    m[i].q = q;
    m[i].bias = q << 22;
    m[i].mu = (uint64_t)((((u128)1) << 64) / q);
  }
  const size_t db_words = (size_t)R * C * kN, sel_words = (size_t)NQ * C * 2 * kN, out_words = (size_t)NQ * R * 2 * kN;
  const size_t dbp_bytes = LAYOUT2 ? (size_t)kN * RT * ((KG + 3) / 4) * L * 1024 : (size_t)kN * RT * KG * L * 256, selp_bytes = (size_t)kN * KG * L * 256;
  uint64_t *db, *sel, *out, *ref;
  uint8_t *dbp, *selp;
  CK(hipMalloc(&db, db_words * 8)); CK(hipMalloc(&sel, sel_words * 8));
  CK(hipMalloc(&out, out_words * 8)); CK(hipMalloc(&ref, out_words * 8));
  CK(hipMalloc(&dbp, dbp_bytes)); CK(hipMalloc(&selp, selp_bytes));
  const int edge = argc > 5 ? atoi(argv[5]) : 0;
  fill_kernel<<<(db_words + 255) / 256, 256>>>(db, db_words, q0, q1, N, kN, 1, edge);
  fill_kernel<<<(sel_words + 255) / 256, 256>>>(sel, sel_words, q0, q1, N, kN, 2, edge);
  CK(hipMemset(out, 0xFF, out_words * 8));
  ref_kernel<<<(out_words + 255) / 256, 256>>>(db, sel, ref, R, C, kN, N, NQ, q0, q1);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  db_pack_kernel<L><<<dim3(kN / 16, RT, KG), 256>>>(db, dbp, R, C, kN, N, RT, KG, q0, q1);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("db_pack: %.3f ms (%.1f MB u64 -> %.1f MB packed)\n", ms, db_words * 8 / 1e6, dbp_bytes / 1e6);
  for (int it = 0; it < 3; ++it) {
    CK(hipEventRecord(e0));
    sel_pack_kernel<L><<<dim3(kN / 16, KG), 256>>>(sel, selp, C, kN, N, KG, NQ, q0, q1);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  printf("sel_pack: %.3f ms (%.1f MB packed)\n", ms, selp_bytes / 1e6);
  scan_mfma_kernel<L, KS><<<kN / 8, 512>>>(dbp, selp, out, RT, KG, R, NQ, kN, N, m[0], m[1]);
  CK(hipDeviceSynchronize());
  std::vector<uint64_t> ho(out_words), hr(out_words);
  CK(hipMemcpy(ho.data(), out, out_words * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hr.data(), ref, out_words * 8, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t i = 0; i < out_words; ++i) if (ho[i] != hr[i]) { if (bad < 5) printf("mismatch @%zu: %llx vs %llx\n", i, (unsigned long long)ho[i], (unsigned long long)hr[i]); ++bad; }
  printf("check: %zu mismatches of %zu\n", bad, out_words);
  float best = 1e9, tot = 0;
  for (int it = 0; it < iters; ++it) {
    CK(hipEventRecord(e0));
    scan_mfma_kernel<L, KS><<<kN / 8, 512>>>(dbp, selp, out, RT, KG, R, NQ, kN, N, m[0], m[1]);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best; tot += ms;
  }
  printf("scan_mfma R=%u C=%u NQ=%u: best %.3f ms avg %.3f ms -> %.2f TB/s on %.1f MB packed (%.2f TB/s u64-equivalent)\n", R, C, NQ,
         best, tot / iters, dbp_bytes / best / 1e9, dbp_bytes / 1e6, db_words * 8 / best / 1e9);
  return bad != 0;
}
