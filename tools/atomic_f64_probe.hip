// atomic_f64_probe.hip -- can the upper level accumulate its products with no-return fp64 atomics in L2?
//
// The split upper level at N = 16384 writes every transformed plaintext (128 KiB of doubles) to scratch and a second
// kernel reads it back to multiply-accumulate it (8.1 GB per query at cfg 5).  The alternative:
// the transform kernel multiplies by the two selector polynomials itself and adds the products into partial sums with
// global_atomic_add_f64 -- sums of <= 16 integer-valued terms below 2^49 are exact in any order.  This probe measures
// what such a stream of atomics costs: WGS workgroups of 1024 threads, each performing REP rounds of 2 x 16 atomics per
// thread (one "transform") into one of REGIONS accumulator pairs (2 x 128 KiB each), the region chosen so that the
// workgroups an XCD runs share few regions (blockIdx % 8 = XCD).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_f64_probe.hip -o tools/atomic_f64_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int NT = 1024, EPT = 16, N = NT * EPT;

template <int MODE>   // 0: atomics, 1: plain stores (the scratch round trip's write half), 2: compute only
__global__ void __launch_bounds__(NT) probe(double* __restrict__ acc, const double* __restrict__ sel, uint32_t regions_per_xcd,
                                            int rep, int spin) {
  const uint32_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  double x[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) x[e] = (double)((threadIdx.x * 31 + e * 7 + blockIdx.x) & 1023);
  for (int r = 0; r < rep; ++r) {
    // stand-in for the transform: `spin` dependent fp64 FMAs per element
    for (int s = 0; s < spin; ++s)
#pragma unroll
      for (int e = 0; e < EPT; ++e) x[e] = __builtin_fma(x[e], 1.0000001, 0.5);
    const uint32_t region = xcd * regions_per_xcd + (slot + r) % regions_per_xcd;
    double* a0 = acc + (size_t)region * 2 * N;
    const double* s0 = sel + (size_t)((slot * 5 + r) % 64) * 2 * N;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const double v = __builtin_trunc(x[e]) * s0[p * N + e * NT + threadIdx.x];
        if (MODE == 0) unsafeAtomicAdd(a0 + p * N + e * NT + threadIdx.x, v);
        else if (MODE == 1) acc[(((size_t)blockIdx.x * rep + r) % 4096) * 2 * N + p * N + e * NT + threadIdx.x] = v;
        else if (v == 1.2345) a0[0] = v;
      }
  }
}

int main(int argc, char** argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 2048, rep = argc > 2 ? atoi(argv[2]) : 12;
  double *acc, *sel;
  const size_t acc_words = (size_t)4096 * 2 * N;    // 1 GiB: room for the plain-store variant's distinct targets
  CHECK(hipMalloc(&acc, acc_words * 8));
  CHECK(hipMalloc(&sel, (size_t)64 * 2 * N * 8));
  CHECK(hipMemset(acc, 0, acc_words * 8));
  CHECK(hipMemset(sel, 0, (size_t)64 * 2 * N * 8));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int spin : {0, 40, 80}) {
    for (uint32_t rpx : {1u, 3u, 12u, 48u}) {
      for (int mode = 0; mode < 3; ++mode) {
        float best = 1e30f;
        for (int it = 0; it < 4; ++it) {
          hipEventRecord(e0);
          if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(wgs), dim3(NT), 0, 0, acc, sel, rpx, rep, spin);
          else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(wgs), dim3(NT), 0, 0, acc, sel, rpx, rep, spin);
          else hipLaunchKernelGGL(probe<2>, dim3(wgs), dim3(NT), 0, 0, acc, sel, rpx, rep, spin);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          if (it && ms < best) best = ms;
        }
        CHECK(hipGetLastError());
        const double transforms = (double)wgs * rep;
        printf("spin %3d  regions/XCD %2u (%5.1f MiB hot per XCD)  %-8s %8.3f ms  %6.2f us per transform per CU  %7.1f GB/s of products\n",
               spin, rpx, rpx * 2.0 * N * 8 / 1048576.0, mode == 0 ? "atomics" : mode == 1 ? "stores" : "compute", best,
               best * 1e3 / (transforms / 256.0), transforms * 2 * N * 8 / (best * 1e6));
      }
    }
  }
  return 0;
}
