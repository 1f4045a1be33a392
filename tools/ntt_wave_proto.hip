// ntt_wave_proto.hip -- experiment (round 3, not product code): is a forward 4096-point negacyclic NTT faster as
//   A  one WORKGROUP of 256 threads per polynomial, 16 residues per thread, three register passes of four stages with two
//      LDS exchanges and barriers (ntt_core.h, what every product kernel uses), or as
//   B  one WAVE per polynomial, 64 residues per lane, two register passes of six stages with ONE exchange and no
//      workgroup barrier, the six high stages on wave-uniform (scalar) twiddles?
// Both load doubles, transform with the same exact fp64 butterflies (arith.h) and store doubles; B leaves its output in
// its own "device order" (slot e * 64 + lane <-> bit-reversed position lane * 64 + e), which would replace A's in a
// product built on it.  Prints per-launch time, transforms per microsecond and checks B against A through the two
// orders.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ntt_wave_proto.hip -o tools/ntt_wave_proto
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../pir_amd/csrc/host_math.h"
#include "../pir_amd/csrc/ntt_core.h"

using namespace pirgpu;

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);        \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

constexpr int LOGN = 12, N = 1 << LOGN, NT = N / 16;

extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

// ---- A: the product's transform
__global__ void __launch_bounds__(NT) ntt_a(const DevParams* __restrict__ P, double* __restrict__ data) {
  const uint32_t tid = threadIdx.x;
  double* poly = data + (size_t)blockIdx.x * N;
  double x[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) x[e] = poly[e * NT + tid];
  ntt_forward<kNttF64, LOGN, true, false>(x, smem_raw, P, 0, tid);
#pragma unroll
  for (int e = 0; e < 16; ++e) poly[e * NT + tid] = x[e];
}

// ---- B: one wave per polynomial
__device__ __forceinline__ void bfly(double& a, double& b, double w, const F64Mod& m) {
  const double t = f64_mulmod(b, w, m);
  b = a - t;
  a = a + t;
}

constexpr int kRow = 65;                       // padded row of the 64 x 64 exchange (doubles)
constexpr int kWaveLds = 64 * kRow;            // doubles per wave

template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES) ntt_b(const DevParams* __restrict__ P, double* __restrict__ data,
                                                     uint32_t npoly) {
  double* lds = reinterpret_cast<double*>(smem_raw);
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t poly = blockIdx.x * WAVES + wave;
  if (poly >= npoly) return;
  double* s = lds + wave * kWaveLds;
  double* p = data + (size_t)poly * N;
  const F64Mod m{P->tab[0].qd, P->tab[0].qinvd, false};
  const double* __restrict__ tw = P->tab[0].twf;   // tw[i] = psi^bitrev(i), centred
  double x[64];
#pragma unroll
  for (int e = 0; e < 64; ++e) x[e] = p[e * 64 + lane];
  // pass 1: stages 0..5 act on the register index (the six high bits of the coefficient index); group g of stage st
  // is the same for every lane -> its twiddle is wave-uniform
#pragma unroll
  for (int st = 0; st < 6; ++st) {
    constexpr int dummy = 0;
    (void)dummy;
    const int half = 32 >> st;
#pragma unroll
    for (int g = 0; g < (1 << st); ++g) {
      const double w = tw[(1 << st) + g];
#pragma unroll
      for (int l = 0; l < half; ++l) bfly(x[g * 2 * half + l], x[g * 2 * half + l + half], w, m);
    }
  }
  // exchange: element e * 64 + lane -> register e' = ... of lane' = e  (a 64 x 64 transpose through padded LDS; no
  // workgroup barrier: a wave's own LDS accesses are ordered by its lgkmcnt waits)
#pragma unroll
  for (int e = 0; e < 64; ++e) s[e * kRow + lane] = x[e];
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int e = 0; e < 64; ++e) x[e] = s[lane * kRow + e];
  // pass 2: lane holds coefficients lane * 64 + e; stages 6..11, per-lane twiddles
#pragma unroll
  for (int st = 6; st < 12; ++st) {
    const int half = 32 >> (st - 6);
    const int ng = 1 << (st - 6);                 // groups inside the lane
    const double* tws = tw + (1 << st) + lane * ng;
#pragma unroll
    for (int g = 0; g < ng; ++g) {
      const double w = tws[g];
#pragma unroll
      for (int l = 0; l < half; ++l) bfly(x[g * 2 * half + l], x[g * 2 * half + l + half], w, m);
    }
  }
#pragma unroll
  for (int e = 0; e < 64; ++e) p[e * 64 + lane] = f64_norm(x[e], m);   // device order of this organisation
}

int main(int argc, char** argv) {
  const uint32_t npoly = argc > 1 ? (uint32_t)atoi(argv[1]) : 16384;
  const uint64_t q = 0xffffee001ull;
  // tables as ctx.hip builds them (fp64 flavour only)
  DevParams hp{};
  hp.N = N;
  hp.logN = LOGN;
  hp.k = 1;
  hp.mod[0].q = q;
  hp.ntt_mode = kNttF64;
  hp.f64_lazy_inv = 1;
  const uint64_t psi = hm::minimal_primitive_root(2ull * N, q);
  std::vector<double> twf(N);
  {
    uint64_t pw = 1;
    for (uint32_t j = 0; j < (uint32_t)N; ++j) {
      const uint32_t r = hm::bitrev(j, LOGN);
      twf[r] = pw > q / 2 ? -(double)(q - pw) : (double)pw;
      pw = hm::mulmod(pw, psi, q);
    }
  }
  double* d_tw;
  CHECK(hipMalloc((void**)&d_tw, N * 8));
  CHECK(hipMemcpy(d_tw, twf.data(), N * 8, hipMemcpyHostToDevice));
  hp.tab[0].twf = d_tw;
  hp.tab[0].qd = (double)q;
  hp.tab[0].qinvd = 1.0 / (double)q;
  DevParams* dp;
  CHECK(hipMalloc((void**)&dp, sizeof(DevParams)));
  CHECK(hipMemcpy(dp, &hp, sizeof(DevParams), hipMemcpyHostToDevice));

  std::vector<double> h((size_t)npoly * N);
  uint64_t sm = 12345;
  for (auto& v : h) {
    sm = sm * 6364136223846793005ull + 1442695040888963407ull;
    v = (double)((sm >> 20) % q);
  }
  double *da, *db;
  CHECK(hipMalloc((void**)&da, h.size() * 8));
  CHECK(hipMalloc((void**)&db, h.size() * 8));
  const size_t lds_a = (size_t)Plan<LOGN>::LDS_WORDS * 8;
  constexpr int WAVES = 4;
  const size_t lds_b = (size_t)WAVES * kWaveLds * 8;
  CHECK(hipFuncSetAttribute((const void*)ntt_a, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
  CHECK(hipFuncSetAttribute((const void*)ntt_b<WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto time_it = [&](auto launch, double* buf, const char* name) -> int {
    CHECK(hipMemcpy(buf, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    launch(buf);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(buf, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    float best = 1e9f, total = 0;
    const int reps = 20;
    for (int r = 0; r < reps; ++r) {   // transforming transformed data again is fine for timing (values stay reduced)
      CHECK(hipEventRecord(e0));
      launch(buf);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
      total += ms;
    }
    printf("%-28s %8.1f us mean %8.1f us min  %6.1f transforms/us  %5.2f TB/s\n", name, total / reps * 1e3, best * 1e3,
           npoly / (best * 1e3), 2.0 * h.size() * 8 / (best * 1e-3) / 1e12);
    return 0;
  };
  auto la = [&](double* buf) { hipLaunchKernelGGL(ntt_a, dim3(npoly), dim3(NT), lds_a, 0, dp, buf); };
  auto lb = [&](double* buf) {
    hipLaunchKernelGGL(ntt_b<WAVES>, dim3((npoly + WAVES - 1) / WAVES), dim3(64 * WAVES), lds_b, 0, dp, buf, npoly);
  };
  if (time_it(la, da, "A 256 threads x 16 (product)")) return 1;
  if (time_it(lb, db, "B 1 wave x 64")) return 1;
  // correctness of B against A: one transform each from the same input
  CHECK(hipMemcpy(da, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(db, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  la(da);
  lb(db);
  CHECK(hipDeviceSynchronize());
  std::vector<double> ra(h.size()), rb(h.size());
  CHECK(hipMemcpy(ra.data(), da, h.size() * 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(rb.data(), db, h.size() * 8, hipMemcpyDeviceToHost));
  // A: SEAL position pos = 16 t + e at slot e * NT + t (signed representative); B: position lane * 64 + e at slot e * 64 + lane
  size_t bad = 0;
  for (uint32_t pl = 0; pl < npoly && pl < 64; ++pl)
    for (uint32_t pos = 0; pos < (uint32_t)N; ++pos) {
      const double va = ra[(size_t)pl * N + (pos & 15) * NT + (pos >> 4)];
      const double vb = rb[(size_t)pl * N + (pos & 63) * 64 + (pos >> 6)];
      auto canon = [&](double v) { return v < 0 ? v + (double)q : v; };
      if (canon(va) != canon(vb)) ++bad;
    }
  printf("B == A on the first polynomials: %s (%zu mismatches)\n", bad ? "NO" : "yes", bad);
  return bad ? 2 : 0;
}
