// tlb_probe.hip -- does the footprint of the database change the bandwidth of the scan's access pattern?
// 256 workgroups x 8 waves; every wave streams contiguous regions of `stream` bytes (16 B per lane per load, 16 loads in
// flight), wave g takes regions g, g + 2048, ... of a buffer of `bytes` bytes.  Prints GB/s for a small and a large
// footprint with the same region size.   hipcc --offload-arch=gfx950 -O3 tools/tlb_probe.hip -o tools/tlb_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t v4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512) probe(const uint8_t* __restrict__ p, size_t n_regions, size_t stream, uint32_t* out) {
  const uint32_t lane = threadIdx.x & 63, g = blockIdx.x * 8 + (threadIdx.x >> 6), total = gridDim.x * 8;
  uint32_t acc = 0;
  for (size_t r = g; r < n_regions; r += total) {
    const uint8_t* base = p + r * stream + lane * 16;
    for (size_t off = 0; off < stream; off += 16 * 1024) {
      v4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = __builtin_nontemporal_load((const v4*)(base + off + (size_t)u * 1024));
#pragma unroll
      for (int u = 0; u < 16; ++u) acc ^= v[u].x + v[u].y + v[u].z + v[u].w;
    }
  }
  if (acc == 0x12345677u) out[0] = acc;
}

int main(int argc, char** argv) {
  const size_t stream = argc > 1 ? (size_t)atol(argv[1]) * 1024 : 384 * 1024;   // KiB per region (multiple of 16)
  const size_t sizes[3] = {(size_t)1280 << 20, (size_t)6 << 30, (size_t)27 << 30};
  uint32_t* out;
  CHECK(hipMalloc(&out, 4));
  for (size_t bytes : sizes) {
    uint8_t* p;
    CHECK(hipMalloc(&p, bytes));
    CHECK(hipMemset(p, 1, bytes));
    const size_t n_regions = bytes / stream;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(a));
      hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, p, n_regions, stream, out);
      CHECK(hipEventRecord(b));
      CHECK(hipEventSynchronize(b));
      float ms;
      CHECK(hipEventElapsedTime(&ms, a, b));
      if (rep == 2) printf("footprint %6.2f GB, %zu KiB regions: %.3f ms = %.0f GB/s\n", bytes / 1e9, stream >> 10, ms, n_regions * stream / ms / 1e6);
    }
    CHECK(hipFree(p));
  }
  return 0;
}
