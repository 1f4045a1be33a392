// ntt_short_proto.hip -- experiment (round 6; not product code): does ONE 4096-point forward
// transform finish sooner on MORE threads?  The narrow expansion levels (<= 16 tree ciphertexts of a lone query) put
// 6 - 96 workgroups on 256 CUs, each level is two DEPENDENT transforms, and a transform by 256 threads x 16 residues
// (three register passes, two LDS exchanges) takes ~5 us of which ~1.3 us is butterfly issue on the CU's four SIMDs.
// Same ntt_core.h code, compiled with 16 / 8 / 4 residues per thread (256 / 512 / 1024 threads: 3 / 4 / 6 passes):
//   for e in 4 3 2; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPIRGPU_LOG_EPT_ALL=$e tools/ntt_short_proto.hip -o tools/ntt_short_proto_$e; done
// Reports, per workgroup count, the time of a CHAIN of dependent launches (what a tree level is) per launch.
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

constexpr int LOGN = 12, N = 1 << LOGN, EPT = Plan<LOGN>::EPT, NT = Plan<LOGN>::NT;

extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

template <bool PF>
__global__ void __launch_bounds__(NT) ntt_k(const DevParams* __restrict__ P, const double* __restrict__ in, double* __restrict__ out) {
  const uint32_t tid = threadIdx.x;
  const double* pi = in + (size_t)blockIdx.x * N;
  double* po = out + (size_t)blockIdx.x * N;
  double x[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) x[e] = pi[e * NT + tid];
  ntt_forward<kNttF64, LOGN, PF, false>(x, smem_raw, P, 0, tid);
#pragma unroll
  for (int e = 0; e < EPT; ++e) po[e * NT + tid] = x[e];
}

int main(int argc, char** argv) {
  const uint64_t q = 0xffffee001ull;
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
  const uint32_t max_poly = 8192;
  std::vector<double> h((size_t)max_poly * N);
  uint64_t sm = 12345;
  for (auto& v : h) {
    sm = sm * 6364136223846793005ull + 1442695040888963407ull;
    v = (double)((sm >> 20) % q);
  }
  double *da, *db;
  CHECK(hipMalloc((void**)&da, h.size() * 8));
  CHECK(hipMalloc((void**)&db, h.size() * 8));
  CHECK(hipMemcpy(da, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  const size_t lds = (size_t)Plan<LOGN>::LDS_WORDS * 8;
  CHECK(hipFuncSetAttribute((const void*)ntt_k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CHECK(hipFuncSetAttribute((const void*)ntt_k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipStream_t st;
  CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("residues per thread %d, threads per transform %d, LDS %zu bytes\n", EPT, NT, lds);
  // checksum of one transform (the three organisations leave different device orders: compare sorted-independent sums)
  hipLaunchKernelGGL(ntt_k<true>, dim3(1), dim3(NT), lds, st, dp, da, db);
  CHECK(hipStreamSynchronize(st));
  {
    std::vector<double> r(N);
    CHECK(hipMemcpy(r.data(), db, N * 8, hipMemcpyDeviceToHost));
    unsigned __int128 s1 = 0, s2 = 0;
    for (double v : r) {
      const uint64_t c = (uint64_t)(v < 0 ? v + (double)q : v) % q;
      s1 += c;
      s2 += (unsigned __int128)c * c;
    }
    printf("checksum (order-independent): sum %llu  sum of squares mod 2^64 %llu\n", (unsigned long long)s1, (unsigned long long)s2);
  }
  const int chain = 40, reps = 25;
  for (int pf = 1; pf >= 0; --pf)
    for (uint32_t wgs : {1u, 6u, 12u, 24u, 48u, 96u, 192u, 768u, 4096u, 8192u}) {
      float best = 1e9f, total = 0;
      for (int r = 0; r < reps + 3; ++r) {
        CHECK(hipEventRecord(e0, st));
        for (int c = 0; c < chain; ++c) {   // dependent: each launch reads what the previous one wrote
          if (pf) hipLaunchKernelGGL(ntt_k<true>, dim3(wgs), dim3(NT), lds, st, dp, c & 1 ? db : da, c & 1 ? da : db);
          else hipLaunchKernelGGL(ntt_k<false>, dim3(wgs), dim3(NT), lds, st, dp, c & 1 ? db : da, c & 1 ? da : db);
        }
        CHECK(hipEventRecord(e1, st));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 3) {
          best = ms < best ? ms : best;
          total += ms;
        }
      }
      printf("EPT %2d  twiddle prefetch %d  workgroups %5u : %7.2f us per dependent launch (mean)  %7.2f (min)\n", EPT, pf, wgs,
             total / reps / chain * 1e3, best / chain * 1e3);
    }
  return 0;
}
