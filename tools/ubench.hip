// ubench.hip -- instruction issue-rate micro-benchmark for the integer / fp64 ops the
// modular arithmetic is built from (gfx950).  Prints lane-ops/s and the rate
// relative to v_fma_f32 (full rate).  Build: hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
constexpr int CHAINS = 8;

#define KERNEL64(NAME, ASM)                                                                  \
  __global__ void NAME(uint64_t* out, uint32_t a, uint32_t b) {                              \
    uint64_t acc[CHAINS];                                                                    \
    uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;                                       \
    for (int c = 0; c < CHAINS; ++c) acc[c] = threadIdx.x + c;                               \
    for (int i = 0; i < ITERS; ++i) {                                                        \
      _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) asm volatile(ASM : "+v"(acc[c]) : "v"(x), "v"(y) : "vcc"); \
    }                                                                                        \
    uint64_t s = 0;                                                                          \
    for (int c = 0; c < CHAINS; ++c) s += acc[c];                                            \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                          \
  }

#define KERNEL32(NAME, ASM)                                                                  \
  __global__ void NAME(uint64_t* out, uint32_t a, uint32_t b) {                              \
    uint32_t acc[CHAINS];                                                                    \
    uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;                                       \
    for (int c = 0; c < CHAINS; ++c) acc[c] = threadIdx.x + c;                               \
    for (int i = 0; i < ITERS; ++i) {                                                        \
      _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) asm volatile(ASM : "+v"(acc[c]) : "v"(x), "v"(y) : "vcc"); \
    }                                                                                        \
    uint32_t s = 0;                                                                          \
    for (int c = 0; c < CHAINS; ++c) s += acc[c];                                            \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                          \
  }

#define KERNELF64(NAME, ASM)                                                                 \
  __global__ void NAME(uint64_t* out, uint32_t a, uint32_t b) {                              \
    double acc[CHAINS];                                                                      \
    double x = 1.0 + 1e-9 * (a + threadIdx.x), y = 1e-9 * (b ^ threadIdx.x);                 \
    for (int c = 0; c < CHAINS; ++c) acc[c] = threadIdx.x + c;                               \
    for (int i = 0; i < ITERS; ++i) {                                                        \
      _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) asm volatile(ASM : "+v"(acc[c]) : "v"(x), "v"(y)); \
    }                                                                                        \
    double s = 0;                                                                            \
    for (int c = 0; c < CHAINS; ++c) s += acc[c];                                            \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;                                \
  }

KERNEL32(k_fma_f32, "v_fma_f32 %0, %1, %2, %0")
KERNEL64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
KERNEL32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %1")
KERNEL32(k_mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0")
KERNEL32(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL32(k_add_co_u32, "v_add_co_u32 %0, vcc, %0, %1")
KERNEL32(k_addc_co_u32, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7")
KERNEL32(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %0")
KERNEL64(k_lshlrev_b64, "v_lshlrev_b64 %0, 3, %0")
KERNEL64(k_cmp_u64, "v_cmp_lt_u64 vcc, %0, %0")
KERNELF64(k_fma_f64, "v_fma_f64 %0, %1, %2, %0")
KERNELF64(k_mul_f64, "v_mul_f64 %0, %0, %1")
KERNELF64(k_add_f64, "v_add_f64 %0, %0, %1")
KERNELF64(k_floor_f64, "v_floor_f64 %0, %0")
KERNELF64(k_rndne_f64, "v_rndne_f64 %0, %0")
__global__ void k_cvt_f64_u32(uint64_t* out, uint32_t a, uint32_t b) {
  double acc[CHAINS];
  uint32_t x = a + threadIdx.x;
  for (int c = 0; c < CHAINS; ++c) acc[c] = c;
  for (int i = 0; i < ITERS; ++i) {
    _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) asm volatile("v_cvt_f64_u32 %0, %1" : "+v"(acc[c]) : "v"(x));
  }
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}
__global__ void k_cvt_u32_f64(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[CHAINS];
  double x = 1.5 * (a + threadIdx.x);
  for (int c = 0; c < CHAINS; ++c) acc[c] = c;
  for (int i = 0; i < ITERS; ++i) {
    _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) asm volatile("v_cvt_u32_f64 %0, %1" : "+v"(acc[c]) : "v"(x));
  }
  uint32_t s = 0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
KERNEL32(k_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
KERNEL32(k_mad_u16x, "v_mad_u32_u16 %0, %1, %2, %0")
KERNEL32(k_dot4_u32_u8, "v_dot4_u32_u8 %0, %1, %2, %0")

typedef void (*kern_t)(uint64_t*, uint32_t, uint32_t);

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s  CUs=%d  clock=%d kHz  LDS/block=%zu  regs/block=%d  L2=%d\n", prop.name,
         prop.multiProcessorCount, prop.clockRate, prop.sharedMemPerBlock, prop.regsPerBlock, prop.l2CacheSize);
  size_t freeb, totalb;
  CHECK(hipMemGetInfo(&freeb, &totalb));
  printf("memory: free %.1f GiB total %.1f GiB\n", freeb / 1073741824.0, totalb / 1073741824.0);
  const int blocks = prop.multiProcessorCount * 8, threads = 256;
  uint64_t* out;
  CHECK(hipMalloc(&out, (size_t)blocks * threads * 8));
  struct { const char* name; kern_t k; } tests[] = {
      {"v_fma_f32", k_fma_f32},           {"v_mad_u64_u32", k_mad_u64_u32},   {"v_mul_lo_u32", k_mul_lo_u32},
      {"v_mul_hi_u32", k_mul_hi_u32},     {"v_mul_u32_u24", k_mul_u32_u24},   {"v_mul_hi_u32_u24", k_mul_hi_u32_u24},
      {"v_mad_u32_u24", k_mad_u32_u24},   {"v_add_u32", k_add_u32},           {"v_add_co_u32", k_add_co_u32},
      {"v_addc_co_u32", k_addc_co_u32},   {"v_alignbit_b32", k_alignbit},     {"v_and_or_b32", k_and_or},
      {"v_cndmask_b32", k_cndmask},       {"v_lshl_add_u64", k_lshl_add_u64}, {"v_lshlrev_b64", k_lshlrev_b64},
      {"v_cmp_lt_u64", k_cmp_u64},        {"v_fma_f64", k_fma_f64},           {"v_mul_f64", k_mul_f64},
      {"v_add_f64", k_add_f64},           {"v_floor_f64", k_floor_f64},       {"v_rndne_f64", k_rndne_f64},
      {"v_cvt_f64_u32", k_cvt_f64_u32},   {"v_cvt_u32_f64", k_cvt_u32_f64},   {"v_pk_mul_lo_u16", k_pk_mul_lo_u16},
      {"v_mad_u32_u16", k_mad_u16x},      {"v_dot4_u32_u8", k_dot4_u32_u8},
  };
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  double base = 0;
  for (auto& t : tests) {
    hipLaunchKernelGGL(t.k, dim3(blocks), dim3(threads), 0, 0, out, 3u, 5u);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(t.k, dim3(blocks), dim3(threads), 0, 0, out, 3u, 5u);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    double ops = (double)blocks * threads * ITERS * CHAINS;
    double rate = ops / (best * 1e-3);
    if (base == 0) base = rate;
    printf("%-18s %8.3f ms  %9.2f Glane-ops/s  rel_to_fma_f32 %.3f  (%.2f slots)\n", t.name, best, rate / 1e9,
           rate / base, base / rate);
  }
  return 0;
}
